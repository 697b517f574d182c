"""Shared fixtures.  `-m "not gpu"` covers the CPU checker (oracle/) against the golden vectors,
the host-side logic and the C-ABI's exported symbols; `-m gpu` tests are the parity tests proper
and drive the HIP kernels through the C-ABI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.pyoracle import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def rtiow_flat():
    from raytrace_amd import scene as S
    return S.flatten(S.rtiow_scene())


@pytest.fixture(scope="session")
def reference():
    """The compiled main.cc, where oracle/_ref/ holds it (it is built from the reference
    checkout by oracle/Makefile and travels to the GPU box as a prebuilt .so)."""
    from oracle.pyoracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libref_main.so not built")
    return Reference()
