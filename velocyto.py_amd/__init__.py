"""velocyto_amd -- MI355X-native core of velocyto.py's post-counting analysis path.

Same call surfaces as the reference (``estimation``, ``neighbors``, ``diffusion``,
``analysis.VelocytoLoom``; velocyto/__init__.py:12-16), every hot loop a hand-written gfx950
HIP kernel behind the C ABI of ``include/velocyto_hip.h``.  Import as ``velocyto_amd``
(see velocyto_amd.py at the repository root).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def build(force: bool = False) -> str:
    """Compile libvelocyto_hip.so for gfx950 (hipcc; works without a GPU)."""
    return _lib.build(force=force)


def __getattr__(name):  # lazy: importing the package must not need torch/GPU
    import importlib
    if name in ("ops", "estimation", "neighbors", "diffusion", "analysis", "speedboosted", "distributed", "loom_io", "serialization", "preprocess", "atlas"):
        return importlib.import_module(f"velocyto_amd.{name}")
    if name in _ROOT_NAMES:                       # the analysis-side names velocyto/__init__.py:12-15 re-exports at the package root
        return getattr(importlib.import_module(f"velocyto_amd.{_ROOT_NAMES[name]}"), name)
    raise AttributeError(name)


_ROOT_NAMES = {"BalancedKNN": "neighbors", "convolve_by_sparse_weights": "neighbors", "fit_slope": "estimation", "_fit1_slope": "estimation",
               "clusters_stats": "estimation", "dump_hdf5": "serialization", "load_hdf5": "serialization", "VelocytoLoom": "analysis",
               "ixs_thatsort_a2b": "analysis", "load_velocyto_hdf5": "analysis"}
