"""Read sharding across the GPUs of one node.

Reads are independent (src/cmain.c:118-120), so the path shards by batch split with no
data-path collective: every rank owns a contiguous range of reads, balanced by CUMULATIVE
SAMPLE COUNT (read lengths vary by 20x in real data), and results are reported in file order.
The only cross-rank traffic is control: a barrier and a MAX over the step time (bench.py), or
a gather of small per-rank summaries.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def partition_by_samples(lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous read ranges [lo, hi) per rank with near-equal cumulative samples.
    Used by `bench.py --scaling strong` (one fixed read population split over the ranks); the C CLI hands whole
    batches to its GPUs round-robin instead (host/sigtk_amd.c: job index % n_gpus), which balances by samples too
    because batches are cut by sample count."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.size
    csum = np.concatenate(([0], np.cumsum(lengths)))
    total = int(csum[-1])
    out, lo = [], 0
    for g in range(world):
        if lo >= n:
            out.append((n, n))
            continue
        hi = n
        if g < world - 1:
            target = total // world * (g + 1)
            hi = lo
            while hi < n and csum[hi + 1] <= target:
                hi += 1
            if hi == lo:
                hi = lo + 1
        out.append((lo, hi))
        lo = hi
    return out


def weak_scaling_slice(rank: int, reads_per_rank: int) -> Tuple[int, int]:
    """bench.py's weak-scaling layout: rank r owns global reads [r*R, (r+1)*R)."""
    return rank * reads_per_rank, (rank + 1) * reads_per_rank


def gather_in_order(local, group=None):
    """All-gather small per-rank python objects and concatenate them in rank (= file) order."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, local, group=group)
    out = []
    for p in parts:
        out.extend(p)
    return out
