#!/usr/bin/env python3
"""bench.py -- headline benchmark: `sigtk event` throughput (raw samples/s, reads/s) on MI355X.

Workload (BASELINE.json configs[1]): 10 000 synthetic DNA reads x 100 000 samples per GPU
(S = 1e9 int16 samples, 2.0 GB), DNA detector parameters, seed 1; inputs and outputs are resident
in HBM when the timed region starts.  One "step" = one pass of the event path (detect -> build
-> exact fallback) over the whole batch.  With --gpus N every rank processes its own batch of the
same size (reads shard embarrassingly; no data-path collective) -> weak scaling.

The oracle is used only in the CPU-baseline leg (rank 0, N=1): as the checker of the benched output and as the
timed baseline.  Prints ONE JSON line on rank 0 (see the driver contract): value = total samples/s over all
ranks; `roofline` = algorithmic HBM bytes of one step (2*S + 16*E + 40*R) over the
HIP-event-measured duration of the path's kernels; `cpu_baseline` = the real reference
(oracle/_ref/libsigtk_ref.so, kind "reference") or the oracle restatement (kind "port") timed
single-threaded on a bounded subsample of the same reads on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=100000)
    ap.add_argument("--rna", type=int, default=0)
    ap.add_argument("--cpu-reads", type=int, default=2000, help="reads in the CPU baseline subsample (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for the barrier / MAX-reduce "
                    "(nccl = RCCL; 'gloo' lets several ranks share one GPU for testing)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sigtk_amd import api, device

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    api.load_library()  # raises if the HIP extension is missing: no fallback
    kind = 1 if args.rna else 0
    first_read = rank * args.reads  # every rank generates a different slice of the read population
    batch = device.synth_reads(args.reads, args.read_len, seed=1, kind=kind, device=dev, first_read=first_read)
    arena = device.EventArena(batch)
    S = batch.total_samples
    R = batch.n_reads

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (untimed)
    for _ in range(max(args.warmup, 1)):
        device.event(batch, arena, args.rna)
    torch.cuda.synchronize()
    st = arena.status()
    E = int(st.n_events_total)
    parity = None
    # ---- timed region: exactly K steps
    L = api.load_library()
    L.sgk_profile_reset()
    L.sgk_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        device.event(batch, arena, args.rna)
    barrier()
    t1 = time.perf_counter()
    L.sgk_profile_enable(0)
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = api.profile_read()
    L.sgk_profile_reset()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = S * world / (elapsed / args.steps)
        # per-step device time of each kernel (HIP events on the stream it was launched on, summed over
        # its launches in a step) and of the whole path ("path:event": first launch -> last kernel done;
        # detector, builder and the rare fallback run back to back on one stream)
        kern = {k: v[0] / args.steps for k, v in prof.items() if not k.startswith("path:")}
        path_ms = prof["path:event"][0] / args.steps if "path:event" in prof else sum(kern.values())
        dominant = max(kern, key=kern.get) if kern else None
        alg_bytes = 2 * S + 16 * E + 40 * R
        achieved = alg_bytes / (path_ms * 1e-3) / 1e9 if path_ms > 0 else 0.0
        traffic = None
        valu = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not args.rna and R == 10000 and args.read_len == 100000:
            # PMC passes of this same command (separate rocprofv3 --pmc runs, see profiles/README.md): HBM bytes
            # per step, and the VALU occupancy that actually bounds the dominant kernel (it is issue-bound: a
            # wave64 vector instruction holds its SIMD for 4 cycles whatever its type)
            try:
                pmc = json.load(open(tpath))
                traffic = pmc.get("hbm_bytes_per_step")
                sq = pmc.get("sq_counters_per_step", {}).get("k_event_detect")
                if sq:
                    busy_ms = sq["SQ_ACTIVE_INST_VALU_quadcycles"] * 4 / 1024 / 2.4e9 * 1e3
                    valu = {"kernel": "k_event_detect", "wave_instructions": sq["SQ_INSTS_VALU"],
                            "busy_ms_at_2.4GHz": round(busy_ms, 2), "source": "profiles/pmc_traffic.json (rocprofv3 --pmc)"}
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes": alg_bytes, "kernels_ms": {k: round(v, 4) for k, v in kern.items()},
                    "dominant_kernel": dominant, "path_ms": round(path_ms, 4), "valu": valu}

        cpu = None
        if args.cpu_reads > 0 and world == 1:
            nb = min(args.cpu_reads, R)
            o_end = int(batch.offsets_host[nb - 1]) + int(batch.lengths_host[nb - 1])
            samples = batch.samples[:o_end].cpu().numpy()
            offs = np.concatenate([batch.offsets_host[:nb], [np.uint64(o_end)]]).astype(np.uint64)
            # the oracle/reference helpers take CSR offsets: reads here are padded to 64 samples, so
            # hand them per-read views through a compacted copy
            lens = batch.lengths_host[:nb].astype(np.int64)
            comp = np.concatenate([samples[int(batch.offsets_host[r]):int(batch.offsets_host[r]) + int(lens[r])]
                                   for r in range(nb)])
            coffs = np.zeros(nb + 1, dtype=np.uint64)
            np.cumsum(lens, out=coffs[1:])
            dig = batch.dig[:nb].cpu().numpy(); off = batch.off[:nb].cpu().numpy(); rng = batch.rng[:nb].cpu().numpy()
            # the CPU leg is the only place the oracle is touched: first as the checker of the benched output
            # (three reads, bit for bit), then as the timed single-thread baseline
            from oracle.oracle import Oracle, RefLib
            orc = Oracle()
            ok = True
            for r in (0, nb // 2, nb - 1):
                o = int(batch.offsets_host[r]); n = int(batch.lengths_host[r])
                raw = batch.samples[o:o + n].cpu().numpy()
                exp = orc.event_raw(raw, float(batch.dig[r]), float(batch.off[r]), float(batch.rng[r]), args.rna)
                got = arena.read_events(r)
                ok &= (got.start.size == exp.start.size and np.array_equal(got.start.astype(np.uint64), exp.start)
                       and np.array_equal(got.mean.view(np.uint32), exp.mean.view(np.uint32))
                       and np.array_equal(got.stdv.view(np.uint32), exp.stdv.view(np.uint32)))
            parity = bool(ok)
            if not ok:
                raise SystemExit("bench: GPU event output differs from the oracle -- the timing above is void")
            try:
                os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})
            except Exception:
                pass
            if RefLib.available():
                ref = RefLib()
                tc0 = time.perf_counter()
                ne = ref.event_batch_count(comp, coffs, dig, off, rng, args.rna)
                tc = time.perf_counter() - tc0
                kind_s = "reference"
            else:
                tc0 = time.perf_counter()
                ne = Oracle().event_batch_count(comp, coffs, dig, off, rng, args.rna, faithful=1)
                tc = time.perf_counter() - tc0
                kind_s = "port"
            cpu = {"value": round(int(lens.sum()) / tc, 1), "unit": "samples/s", "cores": 1, "kind": kind_s,
                   "sample": "first %d reads of the benched batch (%d samples, %d events), %.1f s, 1 thread of %d host cpus"
                             % (nb, int(lens.sum()), ne, tc, os.cpu_count())}

        out = {
            "metric": "event_raw_samples_per_sec",
            "value": round(value, 1),
            "unit": "samples/s",
            "reads_per_sec": round(R * world / (elapsed / args.steps), 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32+f64",
            "data": "synthetic",
            "config": {"workload": "sigtk event (%s params) on %d synthetic reads x %d samples per GPU "
                                   "(BASELINE configs[1]), device-resident" % ("RNA" if args.rna else "DNA", R,
                                                                               args.read_len),
                       "reads_per_gpu": R, "samples_per_read": args.read_len, "events_per_step_rank0": E,
                       "fallback_reads": int(st.n_fallback_reads), "rerun_chunks": int(st.n_rerun_passes),
                       "parallelism": "reads sharded across ranks, no collective"},
            "parity_spot_check": parity,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
