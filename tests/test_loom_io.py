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
    assert back["flag"].dtype == np.bool_ and back["flag"].tolist() == [True, False, True]      # 8-bit enum, as h5py stores bools
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


# ---------------------------------------------------------------------------------------------------------------------------
# Files this repository's writer did NOT produce: tests/golden/loompy_v2.loom / loompy_v3.loom, written by the C program
# tests/golden/make_loom_fixture.c straight on libhdf5 in loompy's layout (chunked (64, 64) gzip-2 /matrix and layers with
# unlimited maxshape, uint16 / uint32 layers, fixed-length ASCII (v2) or variable-length UTF-8 (v3) string attributes,
# LOOM_SPEC_VERSION as a root attribute (v2) or under /attrs (v3)).  The expected numbers are the program's closed formulas.
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NG, NC = 23, 17


def _expected():
    g, c = np.meshgrid(np.arange(NG), np.arange(NC), indexing="ij")
    S = ((g * 7 + c * 3) % 11) * ((g + c) % 3 != 0) + 40000 * ((g == 5) & (c == 4)) + 300 * ((g == 20) & (c == 16))
    U = ((g * 5 + c * 2) % 7) * ((g * c) % 4 != 1)
    A = ((g + 2 * c) % 5 == 0).astype(np.int64)
    return S, U, A


@pytest.mark.parametrize("version", ["v2", "v3"])
def test_read_loompy_layout(loom_io, version):
    path = os.path.join(GOLDEN, f"loompy_{version}.loom")
    layers, ca, ra = loom_io.read_loom(path)
    S, U, A = _expected()
    want = np.uint16 if version == "v2" else np.uint32
    for name, ref in (("spliced", S), ("unspliced", U), ("ambiguous", A)):
        assert layers[name].dtype == want and layers[name].shape == (NG, NC)
        np.testing.assert_array_equal(layers[name], ref)
    assert int(layers["spliced"].max()) == 40000                                  # beyond uint8: the uint16 device path
    genes = [("G\u00e8ne_2" if (version == "v3" and g == 2) else f"Gene_{g * g}") for g in range(NG)]
    assert [str(x) for x in ra["Gene"]] == genes
    assert [str(x) for x in ra["Accession"]] == [f"ENSMUSG{1000 + 37 * g:011d}" for g in range(NG)]
    assert [str(x) for x in ra["Chromosome"]] == [str(1 + g % 19) for g in range(NG)]
    assert [str(x) for x in ra["Strand"]] == ["-" if g % 2 else "+" for g in range(NG)]
    np.testing.assert_array_equal(ra["Start"], 100000 * np.arange(NG) + 17)
    np.testing.assert_array_equal(ra["End"], ra["Start"] + 1500 + 13 * np.arange(NG))
    assert ra["Start"].dtype == np.int64
    assert [str(x) for x in ca["CellID"]] == [f"sample1:{'ABCD'[c % 4]}ACGT{c * 31:04d}x" for c in range(NC)]
    np.testing.assert_array_equal(ca["Clusters"], np.arange(NC) % 3)
    x, y = 0.5 * np.arange(NC) - 3.25, 1.0 / (1 + np.arange(NC))
    if version == "v2":
        np.testing.assert_array_equal(ca["_X"], x)
        np.testing.assert_array_equal(ca["_Y"], y)
    else:
        assert ca["TSNE"].shape == (NC, 2)
        np.testing.assert_array_equal(ca["TSNE"], np.stack([x, y], 1))
    fa = loom_io.read_file_attrs(path)
    assert fa["LOOM_SPEC_VERSION"] == ("2.0.1" if version == "v2" else "3.0.0") and "CreationDate" in fa
    assert loom_io.layer_shape(path, "unspliced") == (NG, NC)
    np.testing.assert_array_equal(loom_io.read_layer_block(path, "spliced", 3, 11), S[:, 3:11])      # hyperslab across the gzip chunks


@pytest.mark.gpu
@pytest.mark.parametrize("version", ["v2", "v3"])
def test_loompy_layout_to_device(loom_io, version):
    """The same files through the facade constructor (analysis.py:56-67) and straight into the CSR form of the atlas path."""
    import velocyto_amd
    from velocyto_amd import ops
    path = os.path.join(GOLDEN, f"loompy_{version}.loom")
    S, U, A = _expected()
    vlm = velocyto_amd.analysis.VelocytoLoom(path, dtype="float64")
    assert np.array_equal(vlm.S, S) and np.array_equal(vlm.U, U) and np.array_equal(vlm.A, A)
    np.testing.assert_array_equal(vlm.initial_cell_size, S.sum(0))
    assert vlm._counts["S"].t.dtype == torch_int16()                # 40 000 does not fit a byte: uint16 bits on the device
    for layer, ref in (("spliced", S), ("unspliced", U)):
        for blk in (5, 64):                                         # blocks smaller and larger than the file
            csr = loom_io.read_layer_csr(path, layer, cell_block=blk)
            assert csr.C == NC and csr.G == NG and csr.nnz == int((ref != 0).sum())
            np.testing.assert_array_equal(csr.to_dense().as_int32().cpu().numpy(), ref.T)
    part = loom_io.read_layer_csr(path, "spliced", cell_block=4, c0=6, c1=15)     # a rank's shard of the cells
    np.testing.assert_array_equal(part.to_dense().as_int32().cpu().numpy(), S.T[6:15])
    np.testing.assert_array_equal(part.row_sums().cpu().numpy(), S.sum(0)[6:15])


def torch_int16():
    import torch
    return torch.int16


def test_array_valued_root_attributes(loom_io, tmp_path):
    """loompy 2 allows array-valued global attributes; H5Aread writes npoints * size bytes, so the reader must size its buffer
    from the attribute's dataspace (a one-element buffer was heap corruption from file input).  The attributes are written
    here straight through libhdf5."""
    import ctypes
    path = str(tmp_path / "attrs.loom")
    loom_io.write_loom(path, {"spliced": np.ones((3, 4), np.uint16), "unspliced": np.ones((3, 4), np.uint16)})
    L = loom_io._lib()
    hid, hs = loom_io.hid_t, loom_io.hsize_t
    L.H5Fopen.restype = hid
    L.H5Acreate2.restype, L.H5Acreate2.argtypes = hid, [hid, ctypes.c_char_p, hid, hid, hid, hid]
    L.H5Awrite.restype, L.H5Awrite.argtypes = ctypes.c_int, [hid, hid, ctypes.c_void_p]
    f = L.H5Fopen(path.encode(), 1, 0)                                  # H5F_ACC_RDWR
    assert f >= 0
    root = L.H5Gopen2(f, b"/", 0)

    def put(name, arr, tp):
        dims = (hs * arr.ndim)(*arr.shape)
        sp = L.H5Screate_simple(arr.ndim, dims, None)
        a = L.H5Acreate2(root, name, tp, sp, 0, 0)
        assert a >= 0 and L.H5Awrite(a, tp, arr.ctypes.data) >= 0
        L.H5Aclose(a); L.H5Sclose(sp)

    vals = np.arange(1000, dtype=np.float64) * 0.5
    ints = np.arange(12, dtype=np.int32).reshape(3, 4)
    put(b"big_float_array", vals, L._native["DOUBLE"])
    put(b"int_matrix", ints, L._native["INT32"])
    st = L.H5Tcopy(L._c_s1)
    L.H5Tset_size(st, 6)
    words = np.array([b"alpha", b"be", b"gamma!"], dtype="S6")
    put(b"fixed_strings", words, st)
    L.H5Tclose(st)
    L.H5Gclose(root); L.H5Fclose(f)
    fa = loom_io.read_file_attrs(path)
    np.testing.assert_array_equal(fa["big_float_array"], vals)
    np.testing.assert_array_equal(fa["int_matrix"], ints)
    assert list(fa["fixed_strings"]) == ["alpha", "be", "gamma!"]


def test_serialization_container_of_any_object(loom_io, tmp_path):
    """serialization.dump_hdf5 (serialization.py:44-97) walks the attributes of ANY object: numeric arrays become datasets of their
    own name, everything else a pickled + zlib-compressed uint8 dataset "&name" (the reference's container format); exclusion,
    compression and pickle protocol are honoured; load_hdf5 (serialization.py:100-115) reads it back into ANY class, made with
    obj_class.__new__ as the reference does.  (A VelocytoLoom rebuilds device state on load: GPU suite.)"""
    import types
    from scipy import sparse
    from velocyto_amd import serialization
    rng = np.random.default_rng(5)
    obj = types.SimpleNamespace(mat=rng.normal(size=(40, 70)), vec=np.arange(9, dtype=np.int64), mask=rng.random(12) > 0.5,
                                names=np.array(["a", "bc", "def"]), ca={"CellID": np.arange(3)}, knn=sparse.random(8, 8, 0.3, format="csr", random_state=1),
                                note="hello", skipped=np.ones(3))
    path = str(tmp_path / "any.hdf5")
    serialization.dump_hdf5(obj, path, data_compression=4, chunks=(16, 16), pickle_protocol=4, exclude_attributes=["skipped"])
    raw = loom_io.hdf5_load(path)
    assert set(raw) == {"mat", "vec", "mask", "&names", "&ca", "&knn", "&note"}
    np.testing.assert_array_equal(raw["mat"], obj.mat)
    np.testing.assert_array_equal(raw["vec"], obj.vec)
    assert raw["mask"].dtype == np.bool_ and np.array_equal(raw["mask"], obj.mask)
    assert raw["&note"].dtype == np.uint8 and serialization._uint2obj(raw["&note"]) == "hello"
    assert list(serialization._uint2obj(raw["&names"])) == ["a", "bc", "def"]
    assert (serialization._uint2obj(raw["&knn"]) != obj.knn).nnz == 0
    assert np.array_equal(serialization._uint2obj(raw["&ca"])["CellID"], np.arange(3))
    assert serialization._uint2obj(serialization._obj2uint({"x": 1}, compression=1, protocol=2)) == {"x": 1}

    class Plain:                                         # no usable __init__: the reference never calls it either
        def __init__(self, required):
            raise AssertionError("load_hdf5 must not call __init__")
    back = serialization.load_hdf5(path, Plain)
    assert isinstance(back, Plain) and back.note == "hello" and not hasattr(back, "skipped")
    np.testing.assert_array_equal(back.mat, obj.mat)
    assert (back.knn != obj.knn).nnz == 0 and list(back.names) == ["a", "bc", "def"]
    with pytest.raises(TypeError):
        serialization.load_hdf5(path, "VelocytoLoom")
