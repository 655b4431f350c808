"""CPU: sgk_stat_plan -- which implementation and which long-read threshold a stat / jnn / prefix call takes for a batch
(host arithmetic of the library; the kernels behind it are tested on the GPU by tests/test_gpu_stat*.py)."""
from sigtk_amd import api


def plan(tool, n_reads, read_len, longest=None, **opt):
    longest = longest or read_len
    o = api.StatOptions(opt.get("kernels", 0), opt.get("long_min", 0))
    return api.stat_plan(tool, n_reads, n_reads * read_len + (longest - read_len), longest, o)


def test_uniform_batches_have_no_long_reads_and_take_the_wave_kernels():
    for tool in ("stat_pa", "jnn", "prefix"):
        p = plan(tool, 125000, 100000)
        assert (p.kernels, p.long_min, p.long_max_reads) == (2, 0, 0)
        assert p.workspace_bytes > 125000 * 4
    assert plan("stat", 60000, 100000).kernels == 2
    assert plan("stat", 125000, 100000).kernels == 1        # plain stat on >= 81 920 reads: the lane kernels
    assert plan("stat", 82000, 100000).kernels == 1
    assert plan("stat_pa", 400000, 5000).kernels == 1


def test_large_batches_of_short_similar_reads_take_the_lane_kernels():
    assert plan("stat", 400000, 5000).kernels == 1
    assert plan("stat", 400000, 5000, kernels=2).kernels == 2
    # round 5: one line per tool in (reads, samples per read) -- stat: max_read_len <= 1.15 n_reads - 5 500 from 8 192 reads
    # on (LANE_RULES, csrc/stat_args.h; the sweep behind it: profiles/r05_lane_vs_wave_sweep.jsonl)
    assert plan("stat", 40000, 5000).kernels == 1
    assert plan("stat", 40000, 20000).kernels == 1           # (32 768 x 16 384: 0.76 of the wave kernels' time)
    assert plan("stat", 40000, 60000).kernels == 2
    assert plan("stat", 10000, 5000).kernels == 1
    assert plan("stat", 10000, 9000).kernels == 2
    assert plan("stat", 5000, 4000).kernels == 2             # under 5 248 reads the wave kernels ...
    assert plan("stat", 5000, 2000).kernels == 1             # ... except for reads of at most 2 048 samples, from 1 024 reads on
    assert plan("stat", 100000, 20000).kernels == 1
    assert plan("stat", 80000, 40000).kernels == 1           # (81 920 x 49 152: 0.76)
    assert plan("stat", 60000, 80000).kernels == 2
    assert plan("stat_pa", 100000, 40000).kernels == 1       # stat + pA: <= 0.75 n_reads - 1 000, at most 49 152
    assert plan("stat_pa", 100000, 60000).kernels == 2
    assert plan("stat", 400000, 5000, longest=16000).kernels == 2   # not of similar length: the longest is 3.2 x the mean
    assert plan("jnn", 400000, 5000).kernels == 1            # jnn: from 16 384 reads on, <= 0.13 n_reads, at most 15 000
    assert plan("jnn", 60000, 5000).kernels == 1             # (49 152 x 4 096: 0.64)
    assert plan("jnn", 60000, 9000).kernels == 2
    assert plan("jnn", 10000, 1000).kernels == 1
    assert plan("jnn", 3000, 300).kernels == 2
    assert plan("jnn", 100000, 16384).kernels == 2
    assert plan("prefix", 400000, 5000).kernels == 2         # prefix: the wave finders at every shape
    assert plan("prefix", 400000, 5000, kernels=1).kernels == 1
    assert plan("jnn", 10, 100000, kernels=1).kernels == 1
    assert plan("jnn", 10, 3000001, kernels=1).long_min == 0  # the long-read path belongs to the wave kernels


def test_the_threshold_follows_the_batch():
    # one 3 000 001-sample read among 20 000 of 100 000: n_samples / 2048 (jnn: / 3072)
    p = plan("stat", 20000, 100000, longest=3000001)
    assert p.kernels == 2 and p.long_min == (20000 * 100000 + 2900001) // 2048 and p.long_max_reads == 128
    assert plan("jnn", 20000, 100000, longest=3000001).long_min == (20000 * 100000 + 2900001) // 3072
    assert plan("prefix", 20000, 100000, longest=3000001).long_min == p.long_min
    # ... among 125 000 it is not long (6.1e6), among 2 000 the floor of the default applies
    assert plan("stat", 125000, 100000, longest=3000001).long_min == 0
    # (round 5: the floor follows the batch and the tool, stat_args.h LongRule -- stat clamp(n_samples / 1024, 131 072,
    # 262 144), jnn 131 072, prefix clamp(n_samples / 512, 196 608, 262 144))
    assert plan("stat", 3000, 100000, longest=3000001).long_min == 262144
    assert plan("stat", 2000, 100000, longest=3000001).long_min == (2000 * 100000 + 2900001) // 1024
    assert plan("stat", 1000, 100000, longest=200000).long_min == 131072
    assert plan("jnn", 3000, 100000, longest=200000).long_min == 131072
    assert plan("prefix", 1000, 100000, longest=200000).long_min == 196608
    assert plan("prefix", 1000, 100000, longest=150000).long_min == 0
    assert plan("prefix", 3000, 100000, longest=250000).long_min == 0
    # a single read
    assert plan("stat", 1, 3000001).long_min == 131072
    assert plan("stat", 1, 262143).long_min == 131072
    assert plan("stat", 1, 131071).long_min == 0


def test_explicit_thresholds():
    assert plan("stat", 1000, 50000, long_min=-1).long_min == 0
    p = plan("stat", 1000, 50000, long_min=20000)
    assert (p.long_min, p.long_max_reads) == (20000, 512)
    assert plan("stat", 1000, 50000, long_min=100).long_min == 8192     # the floor of an explicit threshold
    assert plan("stat", 1000, 5000, long_min=8192).long_min == 0        # no read that long


def test_bad_arguments():
    import ctypes as C
    L = api.load_library()
    p = api.StatPlan()
    assert L.sgk_stat_plan(4, 1, 1, 1, None, C.byref(p)) != 0
    assert L.sgk_stat_plan(0, 1, 1, 1, None, None) != 0
    assert L.sgk_stat_plan(0, 10, 1000, 100, None, C.byref(p)) == 0 and p.kernels == 2
