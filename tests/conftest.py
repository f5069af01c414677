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


@pytest.fixture(scope="session")
def reference_kernels(oracle):
    """The reference's own compiled kernels (oracle/_ref).  Where this tree should have them (a build happened, or /root/reference is
    here to build from) their absence FAILS the test that asks for them; a tree that never had them skips, and says why."""
    st, what = oracle.reference_module_status()
    if st == "present":
        return what
    if oracle.reference_module_expected():
        pytest.fail(f"the reference kernels are expected here but {st}: {what}")
    pytest.skip(f"reference kernels {st}: {what}")


@pytest.fixture(autouse=True)
def _release_device_memory():
    """After every test: drop what the caching allocator still holds.  The suite runs in ONE process and some tests need most of the
    device (cfg5 at 1M cells peaks at 136 GB, the 8-rank rehearsal starts eight more processes on the same GPU): memory cached by an
    earlier test must not starve a later one or the subprocesses it launches."""
    yield
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.empty_cache()
