"""pytest configuration: markers, import paths, shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol table (runs without a GPU).
`-m gpu`      : parity tests proper - every call goes through the C-ABI HIP library on cuda:0.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


@pytest.fixture(scope="session")
def oracle():
    import oracle as _oracle  # oracle/oracle.py - the CPU checker (test infrastructure)
    _oracle.build()
    return _oracle
