#!/usr/bin/env python3
"""PCIe-inclusive rate of one pipelined job (sgk_job_*): pinned host staging -> H2D -> (svb-zd decode) -> event
kernels -> D2H into pinned host buffers, for config-2-shaped batches.  Staging is filled once (untimed); the timed
region is sgk_job_submit + sgk_job_wait.   python tools/bench_job.py [--reads 2000] [--read-len 100000]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=2000)
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--jobs", type=int, default=3, help="jobs kept in flight for the overlapped measurement")
    a = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)
    torch.cuda.init()
    from sigtk_amd import api, blow5
    L = api.load_library()
    reads, dig, off, rng = api.synth_reads_host(a.reads, a.read_len, 1, 0)
    S = a.reads * a.read_len
    out = {}
    for fmt in ("int16", "svbzd"):
        job = api.Job(0)
        sig = [blow5.svb_zd_encode(r) for r in reads] if fmt == "svbzd" else reads
        counts = [r.size for r in reads] if fmt == "svbzd" else None
        in_bytes = sum(len(s) for s in sig) if fmt == "svbzd" else 2 * S
        job.stage(sig, dig, off, rng, counts)
        for flags, name in ((0, "event"), (api.JOB_EVENTS_COMPACT, "event -c")):
            job.launch(api.TOOL_EVENT, flags=flags)   # warm-up: device and result buffers grow
            api.check(L.sgk_job_wait(job.h))
            ts = []
            for _ in range(a.steps):
                t0 = time.perf_counter()
                job.launch(api.TOOL_EVENT, flags=flags)
                api.check(L.sgk_job_wait(job.h))
                ts.append(time.perf_counter() - t0)
            dt = min(ts)
            out["%s / %s" % (fmt, name)] = {"submit_to_done_ms": round(dt * 1e3, 2), "samples_per_s": round(S / dt, 1),
                                            "input_MB": round(in_bytes / 1e6, 1)}
        # several jobs in flight on their own streams: transfers of one overlap the kernels / transfers of the others
        jobs = [job] + [api.Job(0) for _ in range(a.jobs - 1)]
        for j in jobs[1:]:
            j.stage(sig, dig, off, rng, counts)
        for flags, name in ((api.JOB_EVENTS_COMPACT, "event -c"),):
            for j in jobs:
                j.launch(api.TOOL_EVENT, flags=flags)
            for j in jobs:
                api.check(L.sgk_job_wait(j.h))
            rounds = 4
            t0 = time.perf_counter()
            for j in jobs:
                j.launch(api.TOOL_EVENT, flags=flags)
            for _ in range(rounds - 1):
                for j in jobs:
                    api.check(L.sgk_job_wait(j.h))
                    j.launch(api.TOOL_EVENT, flags=flags)
            for j in jobs:
                api.check(L.sgk_job_wait(j.h))
            dt = time.perf_counter() - t0
            out["%s / %s, %d jobs in flight" % (fmt, name, a.jobs)] = {
                "samples_per_s": round(S * rounds * len(jobs) / dt, 1), "ms_per_job": round(dt / (rounds * len(jobs)) * 1e3, 2)}
        for j in jobs:
            j.close()
    print(json.dumps({"reads": a.reads, "samples": S, "results": out}))


if __name__ == "__main__":
    main()
