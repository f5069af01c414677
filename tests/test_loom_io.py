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


def test_hdf5_flat_dump_load(loom_io, tmp_path):
    import pickle, zlib
    rng = np.random.default_rng(2)
    d = {"Sx": rng.normal(size=(7, 9)), "gammas": rng.random(7).astype(np.float32), "ix": np.arange(5, dtype=np.int64),
         "flag": np.array([True, False, True]), "scalar": np.float64(3.5),
         "&ca": np.frombuffer(zlib.compress(pickle.dumps({"CellID": list(range(9))}, protocol=2), 9), dtype=np.uint8)}
    path = str(tmp_path / "ck.hdf5")
    loom_io.hdf5_dump(path, d)
    back = loom_io.hdf5_load(path)
    assert set(back) == set(d)
    np.testing.assert_array_equal(back["Sx"], d["Sx"])
    assert back["gammas"].dtype == np.float32 and back["ix"].dtype == np.int64
    np.testing.assert_array_equal(back["flag"], [1, 0, 1])
    assert back["scalar"].shape == (1,) and back["scalar"][0] == 3.5
    assert pickle.loads(zlib.decompress(back["&ca"].tobytes())) == {"CellID": list(range(9))}


@pytest.mark.gpu
def test_velocytoloom_checkpoint_roundtrip(loom_io, tmp_path):
    import velocyto_amd
    from velocyto_amd.analysis import load_velocyto_hdf5
    rng = np.random.default_rng(3)
    G, C = 30, 48
    vlm = velocyto_amd.analysis.VelocytoLoom.from_arrays(rng.poisson(3, (G, C)).astype(np.uint16), rng.poisson(1, (G, C)).astype(np.uint16), dtype="float64")
    vlm.normalize("both"); vlm.pcs = rng.normal(size=(C, 5)); vlm.knn_imputation(k=5, n_jobs=1); vlm.fit_gammas(fit_offset=False, weighted=False)
    path = str(tmp_path / "vlm.hdf5")
    vlm.to_hdf5(path)
    v2 = load_velocyto_hdf5(path, dtype="float64")
    for name in ("S", "U", "S_sz", "Sx", "Ux_sz"):
        np.testing.assert_array_equal(getattr(v2, name), getattr(vlm, name))
    np.testing.assert_array_equal(v2.gammas, vlm.gammas)
    assert (v2.knn != vlm.knn).nnz == 0 and list(v2.ca) == list(vlm.ca)
    v2.predict_U(); v2.calculate_velocity()                      # the restored object keeps working on the device
    vlm.predict_U(); vlm.calculate_velocity()
    np.testing.assert_array_equal(v2.velocity, vlm.velocity)
