#!/usr/bin/env python3
"""End-to-end CLI comparison on the GPU box: sigtk-amd vs the real reference binary
(oracle/_ref/sigtk_ref) on a synthetic BLOW5 -- byte-compare stdout and report wall times.
    python tests/e2e_compare.py [--reads 500] [--read-len 100000] [--kind 0]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sigtk_amd import api, blow5, build  # noqa: E402
from oracle.oracle import REF_BIN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=500)
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--kind", type=int, default=0)
    ap.add_argument("--cli-args", default="", help="extra options for sigtk-amd, e.g. '--host-decode -t 8'")
    ap.add_argument("--no-ref", type=int, default=0, help="1: time sigtk-amd only (large inputs)")
    ap.add_argument("--ragged", type=int, default=0, help="seed: random read lengths in [300, 2*read_len] instead of a fixed one")
    a = ap.parse_args()
    lens = a.read_len
    if a.ragged:
        import numpy as np
        rs = np.random.RandomState(a.ragged)
        lens = [int(x) for x in np.exp(rs.uniform(np.log(300), np.log(2 * a.read_len), size=a.reads))]
    reads, dig, off, rng = api.synth_reads_host(a.reads, lens, 77, a.kind)
    recs = [blow5.Read("synth-%08d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i])
            for i in range(a.reads)]
    attrs = {"experiment_type": "rna" if a.kind else "genomic_dna",
             "sequencing_kit": "sqk-rna002" if a.kind else "sqk-lsk109"}
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "e2e.blow5")
        blow5.write_blow5(f, recs, attrs)
        out = {"reads": a.reads, "samples": int(sum(len(r) for r in reads)), "file_mb": round(os.path.getsize(f) / 1e6, 1)}
        tools = (["event", "-c"], ["stat"], ["jnn"]) if a.no_ref else \
            (["event", "-c"], ["event"], ["stat"], ["jnn"], ["prefix", "--print-stat"], ["ent"]) + \
            ((["pa"],) if a.reads <= 1000 else ())   # pa prints ~10 bytes per sample
        for tool in tools:
            name = " ".join(tool)
            env = dict(os.environ, SGK_CLI_TIMING="1")
            t0 = time.perf_counter(); g = subprocess.run([build.CLI, *tool, *a.cli_args.split(), f], capture_output=True, env=env); tg = time.perf_counter() - t0
            stages = [ln for ln in g.stderr.decode(errors="replace").splitlines() if ln.startswith("[sigtk-amd]")]
            for ln in stages[1:]:
                print(name, ln, file=sys.stderr)
            if a.no_ref:
                out[name] = {"rc": g.returncode, "sigtk_amd_s": round(tg, 3), "stdout_mb": round(len(g.stdout) / 1e6, 1),
                             "samples_per_s": round(out["samples"] / tg, 1), "stages": stages[0] if stages else None}
                continue
            t0 = time.perf_counter(); r = subprocess.run([REF_BIN, *tool, f], capture_output=True, cwd=tmp); tr = time.perf_counter() - t0
            out[name] = {"identical": g.stdout == r.stdout and g.returncode == 0, "sigtk_amd_s": round(tg, 3),
                         "reference_s": round(tr, 3), "stdout_mb": round(len(r.stdout) / 1e6, 1),
                         "stages": stages[0] if stages else None}
            if r.returncode != 0:
                # the reference died (it does on some inputs, e.g. ragged batches of very short RNA-like reads): compare
                # the rows it completed
                done = r.stdout[:r.stdout.rfind(b"\n") + 1]
                out[name]["reference_rc"] = r.returncode
                out[name]["reference_rows_completed"] = done.count(b"\n")
                out[name]["identical_on_completed_rows"] = g.stdout.startswith(done)
            if g.stdout != r.stdout:   # where: the first rows that differ
                gl, rl = g.stdout.split(b"\n"), r.stdout.split(b"\n")
                diff = [i for i in range(min(len(gl), len(rl))) if gl[i] != rl[i]]
                out[name]["rc"] = g.returncode
                out[name]["rows"] = [len(gl), len(rl)]
                out[name]["differing_rows"] = len(diff)
                out[name]["first_differences"] = [[i, gl[i][:200].decode(errors="replace"), rl[i][:200].decode(errors="replace")]
                                                  for i in diff[:3]]
                out[name]["stderr_tail"] = g.stderr.decode(errors="replace")[-400:]
        if not a.no_ref:
            # qts writes a file: compare what a reader sees in the two outputs
            og, orf = os.path.join(tmp, "g.blow5"), os.path.join(tmp, "r.blow5")
            t0 = time.perf_counter(); g = subprocess.run([build.CLI, "qts", f, "-o", og, "-b", "2"], capture_output=True); tg = time.perf_counter() - t0
            t0 = time.perf_counter(); r = subprocess.run([REF_BIN, "qts", f, "-o", orf, "-b", "2"], capture_output=True, cwd=tmp); tr = time.perf_counter() - t0
            out["qts -b 2"] = {"identical": g.returncode == 0 and r.returncode == 0 and blow5.digest(og) == blow5.digest(orf),
                               "sigtk_amd_s": round(tg, 3), "reference_s": round(tr, 3),
                               "out_mb": round(os.path.getsize(og) / 1e6, 1), "ref_out_mb": round(os.path.getsize(orf) / 1e6, 1)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
