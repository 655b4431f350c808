"""CPU, world_size 2 over gloo: the N>1 path.  Reads shard by contiguous ranges with no data-path
collective; the per-rank results, gathered in rank order, must equal the single-process result.
(The oracle stands in for the GPU kernels here -- there is no GPU in this container; what is
under test is the partitioning / generator slicing / in-order gather.)"""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from sigtk_amd import api, shard


def test_partition_by_samples_balanced_and_contiguous():
    rs = np.random.RandomState(0)
    lens = rs.randint(14000, 270000, size=1000)
    for world in (1, 2, 4, 8):
        parts = shard.partition_by_samples(lens, world)
        assert parts[0][0] == 0 and parts[-1][1] == lens.size
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        loads = [lens[a:b].sum() for a, b in parts]
        assert max(loads) - min(loads) <= 2 * lens.max()
    assert shard.partition_by_samples([5, 5], 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard.partition_by_samples([], 2) == [(0, 0), (0, 0)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.oracle import Oracle
    orc = Oracle()
    lo, hi = shard.weak_scaling_slice(rank, R)
    reads, dig, off, rng = api.synth_reads_host(R, 20000, seed=1, kind=0, first_read=lo)
    local = []
    for r, raw in enumerate(reads):
        ev = orc.event_raw(raw, dig[r], off[r], rng[r], 0)
        local.append((lo + r, int(ev.start.size), int(ev.start.sum() % (1 << 31))))
    allr = shard.gather_in_order(local)
    dist.barrier()
    if rank == 0:
        ret.put(allr)
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(oracle):
    R, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    reads, dig, off, rng = api.synth_reads_host(R * world, 20000, seed=1, kind=0)
    exp = []
    for r, raw in enumerate(reads):
        ev = oracle.event_raw(raw, dig[r], off[r], rng[r], 0)
        exp.append((r, int(ev.start.size), int(ev.start.sum() % (1 << 31))))
    assert got == exp
