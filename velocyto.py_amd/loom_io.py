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
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5S_SELECT_SET = 0
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
                ("H5Dvlen_reclaim", ctypes.c_int, [hid_t, hid_t, hid_t, ctypes.c_void_p]),
                ("H5Pcreate", hid_t, [hid_t]), ("H5Pset_chunk", ctypes.c_int, [hid_t, ctypes.c_int, ctypes.POINTER(hsize_t)]),
                ("H5Pset_deflate", ctypes.c_int, [hid_t, ctypes.c_uint]), ("H5Pclose", ctypes.c_int, [hid_t]),
                ("H5Sselect_hyperslab", ctypes.c_int, [hid_t, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
                ("H5Tenum_create", hid_t, [hid_t]), ("H5Tenum_insert", ctypes.c_int, [hid_t, ctypes.c_char_p, ctypes.c_void_p]),
                ("H5Tget_nmembers", ctypes.c_int, [hid_t]),
                ("H5Aget_num_attrs", ctypes.c_int, [hid_t]),
                ("H5Aopen_by_idx", hid_t, [hid_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, hsize_t, hid_t, hid_t]),
                ("H5Aget_name", ctypes.c_ssize_t, [hid_t, ctypes.c_size_t, ctypes.c_char_p]), ("H5Aget_type", hid_t, [hid_t]),
                ("H5Aget_space", hid_t, [hid_t]), ("H5Aread", ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]), ("H5Aclose", ctypes.c_int, [hid_t])):
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
        elif cls == H5T_ENUM and size == 1:
            # h5py stores numpy bool arrays as an 8-bit enum {FALSE = 0, TRUE = 1} (what the reference's dump_hdf5 writes for
            # boolean attributes such as cv_mean_selected; serialization.py:44-92): read the bytes through the file's own type
            raw = np.empty(shape, dtype=np.int8)
            _check(L.H5Dread(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.ctypes.data), f"read {name}")
            L.H5Tclose(tp)
            L.H5Sclose(sp)
            return raw.astype(np.bool_)
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


def read_file_attrs(path: str) -> Dict[str, object]:
    """File-level attributes of a .loom file: the datasets under /attrs (LOOM_SPEC_VERSION >= 3.0.0) and the HDF5 attributes
    of the root group (loompy 2 wrote LOOM_SPEC_VERSION, CreationDate, ... there).  Values as str / numpy scalars or arrays."""
    L = _lib()
    f = _check(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
    out: Dict[str, object] = {}
    try:
        root = _check(L.H5Gopen2(f, b"/", H5P_DEFAULT), "open root group")
        n = L.H5Aget_num_attrs(root)
        for i in range(max(n, 0)):
            a = _check(L.H5Aopen_by_idx(root, b".", H5_INDEX_NAME, H5_ITER_INC, i, H5P_DEFAULT, H5P_DEFAULT), "open attribute")
            ln = L.H5Aget_name(a, 0, None)
            nb = ctypes.create_string_buffer(ln + 1)
            L.H5Aget_name(a, ln + 1, nb)
            tp = L.H5Aget_type(a)
            cls, size = L.H5Tget_class(tp), L.H5Tget_size(tp)
            # the attribute's dataspace decides the buffer: loompy 2 allows array-valued global attributes, and H5Aread writes
            # npoints * size bytes whatever the caller allocated
            sp = L.H5Aget_space(a)
            nd = L.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sp, dims, None)
            shape = tuple(int(dims[j]) for j in range(max(nd, 0)))
            npts = int(np.prod(shape)) if shape else 1
            key = nb.value.decode()
            if cls == H5T_STRING and npts > 0:
                if L.H5Tis_variable_str(tp) > 0:
                    buf = (ctypes.c_char_p * npts)()
                    L.H5Aread(a, tp, buf)
                    vals = [(buf[j] or b"").decode("utf-8", "replace") for j in range(npts)]
                    L.H5Dvlen_reclaim(tp, sp, H5P_DEFAULT, buf)          # the library allocated the strings: hand them back
                else:
                    raw = ctypes.create_string_buffer(size * npts + 1)
                    L.H5Aread(a, tp, raw)
                    vals = [raw.raw[j * size:(j + 1) * size].split(b"\0", 1)[0].decode("utf-8", "replace") for j in range(npts)]
                out[key] = vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
            elif cls in (H5T_INTEGER, H5T_FLOAT) and npts > 0:
                dt = np.dtype(np.float32 if size == 4 else np.float64) if cls == H5T_FLOAT else np.dtype(("u" if L.H5Tget_sign(tp) == 0 else "i") + str(size))
                v = np.empty(shape, dtype=dt)
                L.H5Aread(a, L._native[_NP2H5[dt]], v.ctypes.data)
                out[key] = v[()] if not shape else v
            L.H5Sclose(sp)
            L.H5Tclose(tp)
            L.H5Aclose(a)
        L.H5Gclose(root)
        g, names = _group_members(L, f, "attrs")
        if g is not None:
            for k in names:
                v = _read_dataset(L, g, k)
                out[k] = v[()] if getattr(v, "shape", None) == () else v
            L.H5Gclose(g)
        return out
    finally:
        L.H5Fclose(f)


def layer_shape(path: str, layer: str = "spliced") -> Tuple[int, int]:
    """(genes, cells) of /layers/<layer> without reading it."""
    L = _lib()
    f = _check(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
    try:
        d = _check(L.H5Dopen2(f, f"layers/{layer}".encode(), H5P_DEFAULT), f"open layer {layer}")
        sp = L.H5Dget_space(d)
        dims = (hsize_t * 2)()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        L.H5Sclose(sp)
        L.H5Dclose(d)
        return int(dims[0]), int(dims[1])
    finally:
        L.H5Fclose(f)


def read_layer_block(path: str, layer: str, c0: int, c1: int) -> np.ndarray:
    """The columns (cells) c0..c1-1 of /layers/<layer>, all genes, as stored: one hyperslab read (chunked / gzip-compressed
    layers are decoded by the HDF5 library)."""
    L = _lib()
    f = _check(L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT), f"open {path}")
    try:
        d = _check(L.H5Dopen2(f, f"layers/{layer}".encode(), H5P_DEFAULT), f"open layer {layer}")
        sp, tp = L.H5Dget_space(d), L.H5Dget_type(d)
        dims = (hsize_t * 2)()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        G, C = int(dims[0]), int(dims[1])
        if not (0 <= c0 < c1 <= C):
            raise IndexError(f"cells {c0}:{c1} outside 0:{C}")
        cls, size = L.H5Tget_class(tp), L.H5Tget_size(tp)
        dt = (np.dtype(np.float32 if size == 4 else np.float64) if cls == H5T_FLOAT else np.dtype(("u" if L.H5Tget_sign(tp) == 0 else "i") + str(size)))
        start, count = (hsize_t * 2)(0, c0), (hsize_t * 2)(G, c1 - c0)
        _check(L.H5Sselect_hyperslab(sp, H5S_SELECT_SET, start, None, count, None), "select hyperslab")
        msp = L.H5Screate_simple(2, count, None)
        out = np.empty((G, c1 - c0), dtype=dt)
        _check(L.H5Dread(d, L._native[_NP2H5[dt]], msp, sp, H5P_DEFAULT, out.ctypes.data), f"read layer {layer}")
        L.H5Sclose(msp); L.H5Sclose(sp); L.H5Tclose(tp); L.H5Dclose(d)
        return out
    finally:
        L.H5Fclose(f)


def read_layer_csr(path: str, layer: str, cell_block: int = 8192, c0: int = 0, c1: Optional[int] = None):
    """A count layer of a .loom file straight into the device CSR form of the atlas path (ops.CsrCounts, cells-major): the
    file is read in blocks of `cell_block` cells (hyperslabs of the genes x cells dataset), each block is transposed and
    compacted on the device, so neither the host nor the device ever holds the dense layer (analysis.py:59-61 loads it
    whole as float64).  c0:c1 restricts to a range of cells (a rank's shard)."""
    import torch
    from . import ops
    dev = ops.require_gpu()
    G, C = layer_shape(path, layer)
    c1 = C if c1 is None else c1
    ptr, idx, dat = [torch.zeros(1, dtype=torch.int64, device=dev)], [], []
    base, wide = 0, False
    for s in range(c0, c1, cell_block):
        e = min(c1, s + cell_block)
        blk = read_layer_block(path, layer, s, e)
        if blk.dtype.kind == "f":
            if not np.array_equal(blk, np.rint(blk)):
                raise ValueError(f"layer {layer} holds non-integer values: not a count layer")
        if blk.size and (blk.min() < 0 or blk.max() > 65535):
            raise ValueError(f"layer {layer}: counts outside 0..65535 cannot be held as uint16")
        t = torch.from_numpy(np.ascontiguousarray(blk.astype(np.int32))).to(dev).t().contiguous()          # (cells, genes)
        nz = t != 0
        cnt = nz.sum(1)
        ptr.append(base + torch.cumsum(cnt, 0))
        base += int(cnt.sum())
        idx.append(torch.nonzero(nz, as_tuple=False)[:, 1].to(torch.int32))
        vals = t[nz]
        wide = wide or bool(vals.numel() and int(vals.max()) > 255)
        dat.append(vals)
    cat = lambda xs, dt: torch.cat(xs) if xs else torch.empty(0, dtype=dt, device=dev)
    vals = cat(dat, torch.int32)
    data = vals.to(torch.int16) if wide else vals.to(torch.uint8)                                     # int16 holds the uint16 bit pattern
    return ops.CsrCounts(torch.cat(ptr), cat(idx, torch.int32), data, G)


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
def hdf5_dump(path: str, arrays: Dict[str, np.ndarray], compression: int = 0, chunks: Tuple[int, int] = (2048, 2048)) -> None:
    """compression > 0: 2-d numeric datasets are written chunked (`chunks`, clipped to the shape) with gzip at that level,
    like serialization.dump_hdf5's data_compression / chunks (serialization.py:44-92)."""
    L = _lib()
    f = _check(L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")
    try:
        for name, arr in arrays.items():
            arr = np.asarray(arr)
            if arr.dtype.kind in ("U", "O"):
                arr = np.char.encode(arr.astype(str), "utf-8")
            is_bool = arr.dtype == np.bool_
            if is_bool:
                arr = arr.astype(np.int8)
            arr = np.ascontiguousarray(arr)
            nd = max(arr.ndim, 1)
            shape = arr.shape if arr.ndim else (1,)
            dims = (hsize_t * nd)(*shape)
            sp = L.H5Screate_simple(nd, dims, None)
            if is_bool:
                # the 8-bit enum h5py uses for numpy bools, so that masks come back as bool arrays (and checkpoints stay
                # interchangeable with files written through h5py by the reference)
                tp = _check(L.H5Tenum_create(L._native["INT8"]), "H5Tenum_create")
                for label, val in ((b"FALSE", 0), (b"TRUE", 1)):
                    v = ctypes.c_int8(val)
                    _check(L.H5Tenum_insert(tp, label, ctypes.byref(v)), "H5Tenum_insert")
            elif arr.dtype.kind == "S":
                tp = L.H5Tcopy(L._c_s1)
                L.H5Tset_size(tp, max(arr.dtype.itemsize, 1))
            elif arr.dtype in _NP2H5:
                tp = L._native[_NP2H5[arr.dtype]]
            else:
                raise TypeError(f"{name}: dtype {arr.dtype} cannot be stored")
            dcpl = H5P_DEFAULT
            if compression > 0 and arr.ndim == 2 and arr.size and arr.dtype.kind != "S":
                dcpl = _check(L.H5Pcreate(hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value), "H5Pcreate")
                ch = (hsize_t * 2)(min(int(chunks[0]), arr.shape[0]), min(int(chunks[1]), arr.shape[1]))
                _check(L.H5Pset_chunk(dcpl, 2, ch), "H5Pset_chunk")
                _check(L.H5Pset_deflate(dcpl, min(int(compression), 9)), "H5Pset_deflate")
            d = _check(L.H5Dcreate2(f, name.encode(), tp, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT), f"create {name}")
            if dcpl != H5P_DEFAULT:
                L.H5Pclose(dcpl)
            if arr.size:
                _check(L.H5Dwrite(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data), f"write {name}")
            L.H5Dclose(d)
            L.H5Sclose(sp)
            if arr.dtype.kind == "S" or is_bool:
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
