"""GPU: the N > 1 paths on real devices.  On a 1-GPU box every test here is SKIPPED (never passed): a scaling curve
needs an 8-GPU node, which only the driver has."""
import json
import os
import subprocess
import sys

import pytest

from sigtk_amd import build

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SP1 = os.path.join(ROOT, "tests", "golden", "sp1_dna.blow5")


def _n_gpus(gpu):
    return gpu.device_count()


def _free_port():
    """a rendezvous port nobody holds right now (a fixed number collides on a shared box)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("tool", ["event", "stat", "prefix"])
def test_cli_two_gpus_equals_one(gpu, tool):
    """sigtk-amd --gpus 2 with small batches (so both devices get work, jobs created lazily on the reader thread,
    hipSetDevice per call, pinned buffers per device) prints the bytes --gpus 1 prints."""
    if _n_gpus(gpu) < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % _n_gpus(gpu))
    args = [tool, "-c", SP1] if tool == "event" else [tool, SP1]
    one = subprocess.run([build.CLI, *args, "--gpus", "1", "--batch-samples", "40000"], capture_output=True, timeout=600)
    two = subprocess.run([build.CLI, *args, "--gpus", "2", "--batch-samples", "40000"], capture_output=True, timeout=600)
    assert one.returncode == 0 and two.returncode == 0, two.stderr.decode()[-1000:]
    assert one.stdout == two.stdout and len(one.stdout) > 1000


def test_bench_two_ranks_rccl(gpu):
    """bench.py as the driver launches it for N = 2: one process per GPU over RCCL, ONE JSON line from rank 0 whose
    value is the sum over both ranks."""
    if _n_gpus(gpu) < 2:
        pytest.skip("needs 2 GPUs (this box has %d)" % _n_gpus(gpu))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "3", "--warmup", "1", "--reads", "2000", "--cpu-reads", "0"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["roofline"]["traffic"] is None and d["roofline"]["recorded_pmc"] is None
    assert d["config"]["samples_per_gpu"] == 2000 * 100000


def test_bench_two_ranks_share_one_gpu_gloo(gpu):
    """The same launch with the gloo backend (both ranks on device 0): exercises the N > 1 code of bench.py --
    barrier, MAX over the step time, SUM of the units, strong-scaling partition -- on a 1-GPU box."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "1000", "--cpu-reads", "0",
                        "--backend", "gloo", "--scaling", "strong", "--ragged", "0.8"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    # one population of 1000 reads was split: rank 0 holds about half of the samples, the value counts all of them
    total = d["value"] * d["ms_per_step"] * 1e-3
    assert 0.3 * total < d["config"]["samples_per_gpu"] < 0.7 * total
    assert d["roofline"]["recorded_pmc"] is None
