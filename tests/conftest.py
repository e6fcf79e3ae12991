import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'cosyvoice2-eu_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _host_cores():
    """CPU share of this process (cgroup quota, else the affinity mask).  A GPU box exposes 256 logical CPUs but grants ~16 to one
    container: torch's default of one thread per visible CPU makes the CPU oracle ~30x slower there (measured: 588 s for a test that
    takes 20 s), so the thread count is pinned to what the container really has."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    try:
        import torch
        torch.set_num_threads(_host_cores())
    except Exception:
        pass


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load
