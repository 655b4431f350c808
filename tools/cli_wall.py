#!/usr/bin/env python3
"""CLI wall times on the GPU box: every subtool RUNS times over one synthetic BLOW5 (default 4 000 x 100 000-sample DNA
reads, 4e8 samples), median / min / max of the wall and of the stages the CLI reports (SGK_CLI_TIMING=1); the reference
binary once per subtool beside it (byte-compare of stdout).  One committed number per subtool = the median.
    python tools/cli_wall.py [--reads 4000] [--runs 9] > profiles/<round>_cli_wall.json"""
import argparse
import json
import os
import re
import statistics
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
    ap.add_argument("--reads", type=int, default=4000)
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--runs", type=int, default=9)
    ap.add_argument("--ref", type=int, default=1)
    a = ap.parse_args()
    reads, dig, off, rng = api.synth_reads_host(a.reads, a.read_len, 77, 0)
    recs = [blow5.Read("synth-%08d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i])
            for i in range(a.reads)]
    out = {"reads": a.reads, "samples": int(sum(len(r) for r in reads)), "runs": a.runs, "host_cpus": os.cpu_count()}
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "wall.blow5")
        blow5.write_blow5(f, recs, {"experiment_type": "genomic_dna", "sequencing_kit": "sqk-lsk109"})
        out["file_mb"] = round(os.path.getsize(f) / 1e6, 1)
        env = dict(os.environ, SGK_CLI_TIMING="1")
        for tool in (["event", "-c"], ["stat"], ["jnn"], ["prefix", "--print-stat"], ["ent"]):
            walls, stages, first = [], {}, None
            for _ in range(a.runs):
                t0 = time.perf_counter()
                g = subprocess.run([build.CLI, *tool, f], capture_output=True, env=env)
                walls.append(time.perf_counter() - t0)
                if g.returncode != 0:
                    raise SystemExit("%s failed: %s" % (tool, g.stderr[-400:]))
                first = first if first is not None else g.stdout
                assert g.stdout == first, "output differs between runs"
                for ln in g.stderr.decode(errors="replace").splitlines():
                    if ln.startswith("[sigtk-amd]") and "HIP init" in ln:
                        for name, v in re.findall(r"([A-Za-z+\- ]+?) ([0-9.]+) s", ln.split(":", 1)[1]):
                            stages.setdefault(name.strip(" |,"), []).append(float(v))
            rec = {"wall_s": {"median": round(statistics.median(walls), 3), "min": round(min(walls), 3),
                              "max": round(max(walls), 3)},
                   "stages_median_s": {k: round(statistics.median(v), 3) for k, v in stages.items()},
                   "hip_init_plus_job_create_s": {"median": round(statistics.median(
                       [x + y for x, y in zip(stages.get("HIP init", [0]), stages.get("job create", [0]))]), 3)},
                   "stdout_mb": round(len(first) / 1e6, 1)}
            if a.ref and os.path.exists(REF_BIN):
                t0 = time.perf_counter()
                r = subprocess.run([REF_BIN, *tool, f], capture_output=True, cwd=tmp)
                rec["reference_s"] = round(time.perf_counter() - t0, 3)
                rec["identical"] = r.stdout == first
            out[" ".join(tool)] = rec
            print(" ".join(tool), rec, file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
