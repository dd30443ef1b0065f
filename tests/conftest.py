import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))          # shared case lists (solve_cases.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs many tiny torch ops: on a box that shows hundreds of host cores the default intra-op pool is
    # several times slower than a small one; 4 threads are as fast as 8 here (18.7 s vs 19.9 s for test_oracle_golden.py)
    # and leave cores free -- with all 8 cores taken, any other load made the OpenMP barriers spin (5-7 minute runs)
    import torch
    torch.set_num_threads(max(1, min(4, os.cpu_count() or 1)))
    # Multi-process GPU tests (tests/test_gpu_distributed.py) take their ranks from a fork SERVER that is started here,
    # before anything in this process has touched the GPU: its children are forked from a GPU-clean process and
    # initialise HIP themselves.  (Spawning -- fork + exec -- out of a process that already holds the GPU is what the
    # GPU boxes forbid.)
    import multiprocessing as mp
    try:
        mp.get_context("forkserver")
        from multiprocessing import forkserver
        forkserver.ensure_running()
    except Exception as e:  # pragma: no cover - platform without forkserver
        print("forkserver unavailable:", e)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(params=["pruned", "brute"])
def solver_mode(request, monkeypatch):
    """The reference-golden ladder runs through BOTH searches of the fused loop: the exact pruned one (the product default
    since round 3, k-d-sorted clouds) and the brute-force sweep (houv_amd.solver.PRUNED = False, clouds as given)."""
    from houv_amd import solver
    monkeypatch.setattr(solver, "PRUNED", request.param == "pruned")
    return request.param
