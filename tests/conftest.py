"""pytest configuration: markers, import path, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reflib():
    """The real reference built from source (oracle/_ref); None when it has not been built."""
    from oracle.oracle import RefLib
    if not RefLib.available():
        return None
    return RefLib()


@pytest.fixture(scope="session")
def sp1():
    """The reference's bundled DNA fixture (test/sp1_dna.blow5: 100 reads, 472 511 samples)."""
    from sigtk_amd import blow5
    return blow5.read_blow5(os.path.join(GOLDEN, "sp1_dna.blow5"))


@pytest.fixture(scope="session")
def gpu():
    """The HIP library on a real GPU.  Fails (does not skip) when it is missing: the GPU tests
    must never pass on a silent fallback."""
    # torch bundles its own HIP runtime: initialise it before libsigtk_gpu.so pulls in /opt/rocm's,
    # otherwise torch.cuda stays unavailable in this process (the reverse order works, as in bench.py)
    import torch
    assert torch.cuda.is_available(), "no GPU visible to torch"
    from sigtk_amd import api
    api.load_library()
    assert api.device_count() > 0, "no GPU visible to libsigtk_gpu.so"
    return api
