""".loom ingest / export without loompy or h5py (analysis.py:56-64 reads the layers
``spliced`` / ``unspliced`` / ``ambiguous`` plus row/col attributes; the writer side is
commands/_run.py:283-297).

A .loom file is plain HDF5: ``/matrix`` (genes x cells), ``/layers/<name>`` (same shape),
``/row_attrs/<name>`` (length genes), ``/col_attrs/<name>`` (length cells).  The image ships the HDF5
C library (libhdf5 1.10, /opt/conda/lib) but no Python binding, so this module binds the dozen C
calls it needs with ctypes.  If h5py is importable it is preferred.

Only what the analysis path needs is implemented: contiguous/chunked/compressed numeric datasets of
rank 1-2 and fixed- or variable-length string attributes.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import numpy as np

_HDF5_CANDIDATES = ("libhdf5.so", "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
                    "/usr/lib/x86_64-linux-gnu/libhdf5.so")
_H5 = None

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5T_VARIABLE = ctypes.c_size_t(-1).value


def _lib():
    global _H5
    if _H5 is None:
        last = None
        for cand in _HDF5_CANDIDATES:
            try:
                L = ctypes.CDLL(cand)
                break
            except OSError as e:
                last = e
        else:
            raise ImportError(f"no HDF5 library found (tried {_HDF5_CANDIDATES}): {last}")
        L.H5open()
        for name, res, args in (
                ("H5Fopen", hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]), ("H5Fcreate", hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
                ("H5Fclose", ctypes.c_int, [hid_t]), ("H5Gopen2", hid_t, [hid_t, ctypes.c_char_p, hid_t]),
                ("H5Gcreate2", hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]), ("H5Gclose", ctypes.c_int, [hid_t]),
                ("H5Dopen2", hid_t, [hid_t, ctypes.c_char_p, hid_t]), ("H5Dclose", ctypes.c_int, [hid_t]),
                ("H5Dget_space", hid_t, [hid_t]), ("H5Dget_type", hid_t, [hid_t]),
                ("H5Dread", ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
                ("H5Dwrite", ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
                ("H5Dcreate2", hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                ("H5Sget_simple_extent_ndims", ctypes.c_int, [hid_t]),
                ("H5Sget_simple_extent_dims", ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
                ("H5Screate_simple", hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
                ("H5Sclose", ctypes.c_int, [hid_t]), ("H5Tget_class", ctypes.c_int, [hid_t]), ("H5Tget_size", ctypes.c_size_t, [hid_t]),
                ("H5Tget_sign", ctypes.c_int, [hid_t]), ("H5Tis_variable_str", ctypes.c_int, [hid_t]), ("H5Tclose", ctypes.c_int, [hid_t]),
                ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", ctypes.c_int, [hid_t, ctypes.c_size_t]),
                ("H5Lexists", ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
                ("H5Gget_num_objs", ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t)]),
                ("H5Lget_name_by_idx", ctypes.c_ssize_t, [hid_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, hsize_t, ctypes.c_char_p, ctypes.c_size_t, hid_t]),
                ("H5Dvlen_reclaim", ctypes.c_int, [hid_t, hid_t, hid_t, ctypes.c_void_p])):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L._native = {k: hid_t.in_dll(L, f"H5T_NATIVE_{k}_g").value for k in
                     ("FLOAT", "DOUBLE", "INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64")}
        L._c_s1 = hid_t.in_dll(L, "H5T_C_S1_g").value
        _H5 = L
    return _H5


_NP2H5 = {np.dtype(k): v for k, v in (("float32", "FLOAT"), ("float64", "DOUBLE"), ("int8", "INT8"), ("uint8", "UINT8"), ("int16", "INT16"),
                                      ("uint16", "UINT16"), ("int32", "INT32"), ("uint32", "UINT32"), ("int64", "INT64"), ("uint64", "UINT64"))}


def _check(v, what):
    if v < 0:
        raise IOError(f"HDF5 call failed: {what}")
    return v


def _read_dataset(L, parent: int, name: str) -> np.ndarray:
    d = _check(L.H5Dopen2(parent, name.encode(), H5P_DEFAULT), f"open dataset {name}")
    try:
        sp, tp = L.H5Dget_space(d), L.H5Dget_type(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(sp, dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        cls, size = L.H5Tget_class(tp), L.H5Tget_size(tp)
        if cls == H5T_FLOAT:
            dt = np.dtype(np.float32 if size == 4 else np.float64)
        elif cls == H5T_INTEGER:
            dt = np.dtype(("u" if L.H5Tget_sign(tp) == 0 else "i") + str(size))
        elif cls == H5T_STRING:
            n = int(np.prod(shape)) if shape else 1
            if L.H5Tis_variable_str(tp) > 0:
                buf = (ctypes.c_char_p * n)()
                _check(L.H5Dread(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), f"read {name}")
                out = np.array([(b or b"").decode("utf-8", "replace") for b in buf], dtype=object).reshape(shape)
                L.H5Dvlen_reclaim(tp, sp, H5P_DEFAULT, buf)
            else:
                raw = np.empty(n, dtype=f"S{size}")
                _check(L.H5Dread(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.ctypes.data), f"read {name}")
                out = np.char.decode(raw, "utf-8").reshape(shape)
            L.H5Tclose(tp)
            L.H5Sclose(sp)
            return out
        else:
            raise IOError(f"dataset {name}: unsupported HDF5 type class {cls}")
        out = np.empty(shape, dtype=dt)
        _check(L.H5Dread(d, L._native[_NP2H5[dt]], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), f"read {name}")
        L.H5Tclose(tp)
        L.H5Sclose(sp)
        return out
    finally:
        L.H5Dclose(d)


def _group_members(L, parent: int, name: str):
    if L.H5Lexists(parent, name.encode(), H5P_DEFAULT) <= 0:
        return None, []
    g = _check(L.H5Gopen2(parent, name.encode(), H5P_DEFAULT), f"open group {name}")
    n = hsize_t()
    L.H5Gget_num_objs(g, ctypes.byref(n))
    names = []
    for i in range(n.value):
        ln = L.H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT)
        buf = ctypes.create_string_buffer(ln + 1)
        L.H5Lget_name_by_idx(g, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, ln + 1, H5P_DEFAULT)
        names.append(buf.value.decode())
    return g, names


def read_loom(path: str) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """-> (layers {spliced, unspliced, ambiguous?}, col_attrs, row_attrs), arrays as stored (genes x cells)."""
    try:
        import h5py
        with h5py.File(path, "r") as f:
            layers = {k: f["layers"][k][:, :] for k in ("spliced", "unspliced", "ambiguous") if k in f["layers"]}
            return layers, {k: f["col_attrs"][k][...] for k in f["col_attrs"]}, {k: f["row_attrs"][k][...] for k in f["row_attrs"]}
    except ImportError:
        pass
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    L = _lib()
    f = _check(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
    try:
        g, names = _group_members(L, f, "layers")
        if g is None or "spliced" not in names or "unspliced" not in names:
            raise IOError(f"{path}: not a velocyto loom file (needs /layers/spliced and /layers/unspliced)")
        layers = {k: _read_dataset(L, g, k) for k in ("spliced", "unspliced", "ambiguous") if k in names}
        L.H5Gclose(g)
        attrs = []
        for grp in ("col_attrs", "row_attrs"):
            g, names = _group_members(L, f, grp)
            attrs.append({k: _read_dataset(L, g, k) for k in names} if g is not None else {})
            if g is not None:
                L.H5Gclose(g)
        return layers, attrs[0], attrs[1]
    finally:
        L.H5Fclose(f)


def write_loom(path: str, layers: Dict[str, np.ndarray], col_attrs: Optional[Dict[str, np.ndarray]] = None,
               row_attrs: Optional[Dict[str, np.ndarray]] = None, matrix: Optional[np.ndarray] = None) -> None:
    """Write the layout velocyto's counting pipeline produces (commands/_run.py:283-297): /matrix float32,
    /layers/<name>, /row_attrs/<name>, /col_attrs/<name> (numeric or fixed-length string arrays)."""
    L = _lib()
    f = _check(L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")

    def put(parent, name, arr):
        arr = np.asarray(arr)
        if arr.dtype.kind in ("U", "O"):
            arr = np.char.encode(arr.astype(str), "utf-8")
        arr = np.ascontiguousarray(arr)
        dims = (hsize_t * arr.ndim)(*arr.shape)
        sp = L.H5Screate_simple(arr.ndim, dims, None)
        if arr.dtype.kind == "S":
            tp = L.H5Tcopy(L._c_s1)
            L.H5Tset_size(tp, max(arr.dtype.itemsize, 1))
        else:
            tp = L._native[_NP2H5[arr.dtype]]
        d = _check(L.H5Dcreate2(parent, name.encode(), tp, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create {name}")
        _check(L.H5Dwrite(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data), f"write {name}")
        L.H5Dclose(d)
        L.H5Sclose(sp)
        if arr.dtype.kind == "S":
            L.H5Tclose(tp)

    try:
        first = next(iter(layers.values()))
        put(f, "matrix", np.asarray(matrix if matrix is not None else sum(np.asarray(v, dtype=np.float32) for v in layers.values()), dtype=np.float32))
        for grp, items in (("layers", layers), ("col_attrs", col_attrs or {"CellID": np.arange(first.shape[1])}),
                           ("row_attrs", row_attrs or {"Gene": np.arange(first.shape[0])})):
            g = _check(L.H5Gcreate2(f, grp.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create group {grp}")
            for k, v in items.items():
                put(g, k, v)
            L.H5Gclose(g)
    finally:
        L.H5Fclose(f)


# ----------------------------------------------------------------------------------------------------------
# Flat HDF5 dump / load of a dict of arrays: the container format of velocyto/serialization.py:44-115
# (every ndarray attribute -> a dataset under its name; anything else -> pickle + zlib -> uint8 dataset "&name").
def hdf5_dump(path: str, arrays: Dict[str, np.ndarray]) -> None:
    L = _lib()
    f = _check(L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")
    try:
        for name, arr in arrays.items():
            arr = np.asarray(arr)
            if arr.dtype.kind in ("U", "O"):
                arr = np.char.encode(arr.astype(str), "utf-8")
            if arr.dtype == np.bool_:
                arr = arr.astype(np.uint8)
            arr = np.ascontiguousarray(arr)
            nd = max(arr.ndim, 1)
            shape = arr.shape if arr.ndim else (1,)
            dims = (hsize_t * nd)(*shape)
            sp = L.H5Screate_simple(nd, dims, None)
            if arr.dtype.kind == "S":
                tp = L.H5Tcopy(L._c_s1)
                L.H5Tset_size(tp, max(arr.dtype.itemsize, 1))
            elif arr.dtype in _NP2H5:
                tp = L._native[_NP2H5[arr.dtype]]
            else:
                raise TypeError(f"{name}: dtype {arr.dtype} cannot be stored")
            d = _check(L.H5Dcreate2(f, name.encode(), tp, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create {name}")
            if arr.size:
                _check(L.H5Dwrite(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data), f"write {name}")
            L.H5Dclose(d)
            L.H5Sclose(sp)
            if arr.dtype.kind == "S":
                L.H5Tclose(tp)
    finally:
        L.H5Fclose(f)


def hdf5_load(path: str) -> Dict[str, np.ndarray]:
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    L = _lib()
    f = _check(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
    try:
        g, names = _group_members(L, f, "/")
        out = {k: _read_dataset(L, f, k) for k in names}
        if g is not None:
            L.H5Gclose(g)
        return out
    finally:
        L.H5Fclose(f)
