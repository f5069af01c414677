"""Round trip of the .loom container (HDF5 through ctypes-bound libhdf5, no h5py/loompy): the layout the
reference's counting pipeline writes (commands/_run.py:283-297) and its analysis constructor reads
(analysis.py:56-64).  CPU-only."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def loom_io():
    import velocyto_amd
    from velocyto_amd import loom_io as m
    try:
        m._lib()
    except ImportError as e:
        pytest.skip(f"no HDF5 C library in this environment: {e}")
    return m


def test_loom_roundtrip(loom_io, tmp_path):
    rng = np.random.default_rng(0)
    G, C = 37, 53
    layers = {"spliced": rng.poisson(3, (G, C)).astype(np.uint16), "unspliced": rng.poisson(1, (G, C)).astype(np.uint16),
              "ambiguous": rng.poisson(0.2, (G, C)).astype(np.uint32)}
    ca = {"CellID": np.array([f"cell_{i:03d}" for i in range(C)]), "Clusters": rng.integers(0, 5, C).astype(np.int64), "_X": rng.normal(size=C)}
    ra = {"Gene": np.array([f"G{i}" for i in range(G)]), "Chromosome": np.array(["chr1", "chrX"] * 18 + ["chr2"]), "Start": np.arange(G, dtype=np.int64)}
    path = str(tmp_path / "t.loom")
    loom_io.write_loom(path, layers, ca, ra)
    assert os.path.getsize(path) > 0
    L, ca2, ra2 = loom_io.read_loom(path)
    for k in layers:
        assert L[k].dtype == layers[k].dtype and np.array_equal(L[k], layers[k])
    assert set(ca2) == set(ca) and set(ra2) == set(ra)
    assert list(ca2["CellID"]) == list(ca["CellID"]) and list(ra2["Chromosome"]) == list(ra["Chromosome"])
    np.testing.assert_array_equal(ca2["Clusters"], ca["Clusters"])
    np.testing.assert_array_equal(ca2["_X"], ca["_X"])
    with pytest.raises(FileNotFoundError):
        loom_io.read_loom(str(tmp_path / "missing.loom"))
    bad = str(tmp_path / "bad.loom")
    loom_io.write_loom(bad, {"other": layers["spliced"]})
    with pytest.raises(IOError):
        loom_io.read_loom(bad)


@pytest.mark.gpu
def test_velocytoloom_from_loom_file(loom_io, tmp_path):
    import velocyto_amd
    rng = np.random.default_rng(1)
    G, C = 40, 64
    S, U = rng.poisson(3, (G, C)).astype(np.uint16), rng.poisson(1, (G, C)).astype(np.uint16)
    path = str(tmp_path / "v.loom")
    loom_io.write_loom(path, {"spliced": S, "unspliced": U, "ambiguous": np.zeros_like(S)}, {"CellID": np.arange(C)}, {"Gene": np.arange(G)})
    vlm = velocyto_amd.analysis.VelocytoLoom(path, dtype="float64")
    assert vlm.loom_filepath == path and np.array_equal(vlm.S, S) and np.array_equal(vlm.U, U) and vlm.A.sum() == 0
    np.testing.assert_array_equal(vlm.initial_cell_size, S.sum(0))
    assert list(vlm.ca) == ["CellID"] and list(vlm.ra) == ["Gene"]
