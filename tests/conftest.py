import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # torch's deprecation notice for the old-style weight norm the reference (and therefore the mirror) uses: test noise only -- the
    # library itself installs no warning filters
    config.addinivalue_line("filterwarnings", "ignore:.*weight_norm.*is deprecated.*:FutureWarning")


def load_golden(name):
    """-> (weights dict keyed like the reference's state_dict, dict of the other arrays)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    w = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    a = {k: z[k] for k in z.files if not k.startswith("w.")}
    return w, a


def usable_cores():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU boxes show 256 cores under a quota of
    16: OpenMP / OpenBLAS defaults oversubscribe 16-fold and the oracle-bound tests take minutes instead of seconds)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


@pytest.fixture(scope="session")
def oracle():
    from oracle import visinger_oracle as orc
    orc.build()
    n = usable_cores()
    orc.set_threads(n)
    try:                                  # numpy's BLAS (the oracle's attention matmuls) likewise
        import threadpoolctl
        orc._blas_limit = threadpoolctl.threadpool_limits(limits=n)
    except Exception:
        pass
    return orc


@pytest.fixture
def vs_option():
    """vs_option(name, value): set a dispatch switch (library: vs_set_option; Python layer: _lib.PY_SWITCHES) for this test only.
    The switches are read from the environment once, at load: tests change them through the C ABI, not through os.environ."""
    from visinger_amd import _lib as L
    old = {}

    def set_(name, value):
        old.setdefault(name, L.get_option(name))
        L.set_option(name, value)

    yield set_
    for name, v in old.items():
        L.set_option(name, v)
