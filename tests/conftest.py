import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "learned-pmctf_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# The oracle's C kernels start one thread per online CPU (up to 64); on a box whose CPU share is smaller than the machine
# (the 1-GPU boxes own 16 CPUs) that oversubscribes.  Size both thread pools to what this process may use.
_ncpu = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
os.environ.setdefault("PM_ORACLE_THREADS", str(_ncpu))
os.environ.setdefault("OMP_NUM_THREADS", str(_ncpu))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built libraries (they are git-ignored): build them once, as __graft_entry__.build() does
    libs = [os.path.join(ROOT, "learned-pmctf_amd", "lib", n) for n in ("libpmctf_hip.so", "libpmctf_rans.so")]
    if not all(os.path.exists(p) for p in libs):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible: the product path has no CPU fallback")
    return torch.device("cuda:0")
