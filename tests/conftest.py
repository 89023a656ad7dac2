import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _granted_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota.  A 1-GPU box shows every core of the host in the mask
    but grants a 16-core share; torch's default (one thread per visible core) then runs the CPU oracle many times slower than 16 threads do - the
    oracle-heavy GPU tests took 4 minutes instead of one (round 5: 826 s for the suite against the driver's 900 s limit)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    return max(1, min(n, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    return max(1, min(n, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 16))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    n = _granted_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(n))          # child processes (script round trips, gloo ranks set their own)
    try:
        import torch
        torch.set_num_threads(n)
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
