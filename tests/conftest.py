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


def load_golden(name):
    """-> (weights dict keyed like the reference's state_dict, dict of the other arrays)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    w = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
    a = {k: z[k] for k in z.files if not k.startswith("w.")}
    return w, a


@pytest.fixture(scope="session")
def oracle():
    from oracle import visinger_oracle as orc
    orc.build()
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
