#!/usr/bin/env python3
"""Steady-state end-to-end rate of the drop-in CLI (VERDICT r04 task 3): every per-record subtool over a BLOW5 large
enough to have a steady state (default 100 000 x 100 000-sample DNA reads = 1e10 samples, zlib records + svb-zd signal,
~8.5 GB, page-cache resident), with the stage sums the CLI reports (SGK_CLI_TIMING=1), and the reference binary on the
base file the large one is made of.

    python tools/cli_steady.py [--copies 25] [--base-reads 4000] [--threads 0] [--dir /tmp] > profiles/r05_cli_steady.json

The large file is `copies` concatenations of the records of a base file (4 000 synthetic reads, written by
sigtk_amd.blow5 as slow5lib would write them): what the CLI must print for it is the header line and `copies` times the
body it prints for the base file -- which is compared, byte for byte, with the reference binary's output on the base file
(the 1 / copies subsample VERDICT asked for), and with an md5 of the large run's stdout."""
import argparse
import hashlib
import json
import os
import re
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sigtk_amd import api, blow5, build  # noqa: E402
from oracle.oracle import REF_BIN  # noqa: E402


def replicate(base, big, copies):
    buf = open(base, "rb").read()
    (hsize,) = struct.unpack_from("<I", buf, 64)
    body0 = 68 + hsize
    assert buf[-5:] == blow5.EOF_MARK
    with open(big, "wb") as fh:
        fh.write(buf[:body0])
        for _ in range(copies):
            fh.write(buf[body0:-5])
        fh.write(blow5.EOF_MARK)
    return os.path.getsize(big)


def stages(stderr):
    out = {}
    for ln in stderr.decode(errors="replace").splitlines():
        if ln.startswith("[sigtk-amd]") and "HIP init" in ln:
            m = re.match(r"\[sigtk-amd\] (\d+) reads, (\d+) samples, (\d+) threads, (\d+) GPU", ln)
            if m:
                out["reads"], out["samples"], out["threads"], out["gpus"] = (int(x) for x in m.groups())
            for name, v in re.findall(r"([A-Za-z+\- ]+?) ([0-9.]+) s", ln.split(":", 1)[1]):
                out[name.strip(" |,")] = float(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=25)
    ap.add_argument("--base-reads", type=int, default=4000)
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--threads", type=int, default=0, help="-t for the CLI (0: its default)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--tools", default="stat;jnn;prefix --print-stat;event -c")
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--extra", default="", help="extra CLI arguments, e.g. '--batch-samples 64000000'")
    a = ap.parse_args()
    base = os.path.join(a.dir, "steady_base_%d.blow5" % a.base_reads)
    big = os.path.join(a.dir, "steady_big_%d_x%d.blow5" % (a.base_reads, a.copies))
    t0 = time.perf_counter()
    if not os.path.exists(base):
        reads, dig, off, rng = api.synth_reads_host(a.base_reads, a.read_len, 77, 0)
        recs = [blow5.Read("synth-%08d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i])
                for i in range(a.base_reads)]
        blow5.write_blow5(base, recs, {"experiment_type": "genomic_dna", "sequencing_kit": "sqk-lsk109"})
    if not os.path.exists(big):
        replicate(base, big, a.copies)
    out = {"base_file_mb": round(os.path.getsize(base) / 1e6, 1), "big_file_mb": round(os.path.getsize(big) / 1e6, 1),
           "copies": a.copies, "reads": a.base_reads * a.copies, "samples": a.base_reads * a.copies * a.read_len,
           "host_cpus": os.cpu_count(), "made_in_s": round(time.perf_counter() - t0, 1), "cli_extra": a.extra, "tools": {}}
    env = dict(os.environ, SGK_CLI_TIMING="1")
    targ = (["-t", str(a.threads)] if a.threads else []) + (["--gpus", str(a.gpus)] if a.gpus != 1 else []) + a.extra.split()
    for tool in [t.split() for t in a.tools.split(";")]:
        rec = {}
        # the base file: ours against the reference, byte for byte
        g = subprocess.run([build.CLI, *tool, *targ, base], capture_output=True, env=env)
        if g.returncode != 0:
            raise SystemExit("%s failed on the base file: %s" % (tool, g.stderr[-400:]))
        nl = g.stdout.index(b"\n") + 1
        head, body = g.stdout[:nl], g.stdout[nl:]
        if os.path.exists(REF_BIN):
            t0 = time.perf_counter()
            r = subprocess.run([REF_BIN, *tool, base], capture_output=True, cwd=a.dir)
            rec["reference_on_base_s"] = round(time.perf_counter() - t0, 2)
            rec["reference_samples_per_s"] = round(a.base_reads * a.read_len / (time.perf_counter() - t0))
            rec["base_identical_to_reference"] = r.stdout == g.stdout
        want = hashlib.md5(head)
        for _ in range(a.copies):
            want.update(body)
        rec["stdout_mb"] = round((len(head) + a.copies * len(body)) / 1e6, 1)
        # the large file: timed to /dev/null, then once more through md5
        walls, st = [], {}
        for _ in range(a.runs):
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as nul:
                p = subprocess.run([build.CLI, *tool, *targ, big], stdout=nul, stderr=subprocess.PIPE, env=env)
            walls.append(time.perf_counter() - t0)
            if p.returncode != 0:
                raise SystemExit("%s failed on the large file: %s" % (tool, p.stderr[-400:]))
            st = stages(p.stderr)
        p = subprocess.Popen([build.CLI, *tool, *targ, big], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        h = hashlib.md5()
        while True:
            chunk = p.stdout.read(1 << 24)
            if not chunk:
                break
            h.update(chunk)
        p.wait()
        rec["large_output_is_copies_x_base"] = h.hexdigest() == want.hexdigest()
        w = min(walls)
        rec["wall_s"] = [round(x, 3) for x in walls]
        rec["samples_per_s"] = round(out["samples"] / w)
        rec["stages_s"] = st
        work = w - st.get("HIP init", 0.0) - st.get("job create", 0.0)
        rec["samples_per_s_without_startup"] = round(out["samples"] / work) if work > 0 else None
        rec["signal_bytes_per_s_over_pcie"] = round(os.path.getsize(big) / w)   # (svb-zd blobs ~ the file's bytes)
        out["tools"][" ".join(tool)] = rec
        print(" ".join(tool), json.dumps(rec), file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
