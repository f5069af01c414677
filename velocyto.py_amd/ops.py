"""Device-level operators: thin, typed wrappers over the C ABI (include/velocyto_hip.h).

PyTorch is plumbing only here: it owns device memory and streams; every computation is a
hand-written gfx950 kernel in libvelocyto_hip.so reached through ctypes with raw device
pointers.  No function in this module has a CPU path.

Matrices live on the device CELLS-MAJOR: ``CellMatrix.t`` has shape ``(C, ld)`` with the
logical gene count ``G <= ld`` (ld = G rounded up to 64 elements so that every cell's gene
vector starts 256-byte aligned).  This is the transpose of the reference's ``(G, C)`` arrays.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

F32, F64, U16, U8 = 0, 1, 2, 3
LINEAR, SQRT, LOG10 = 0, 1, 2
RULES_FULL, RULES_PARTIAL, RULES_PARTIAL_NOPSC = 0, 1, 2
PSC_NEGLIGIBLE = 1e-9       # f32 sqrt: a pseudocount at or below this is a candidate for RULES_PARTIAL_NOPSC ...
RULE_NAMES = {0: "full", 1: "partial (literal, speedboosted.pyx:372-378)", 2: "partial, pseudocount below f32 resolution dropped: sign(t) sqrt|t|"}
SCALE_ORDINARY = 1e-4       # ... on a matrix whose mean |e| is at least this (psc / scale <= 1e-5, see partial_rules_for)
TRANSFORMS = {"linear": LINEAR, "sqrt": SQRT, "log10": LOG10, "log": LOG10}

_DT = {torch.float32: F32, torch.float64: F64}


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("velocyto_amd needs a ROCm GPU (MI355X): there is no CPU fallback for the HIP path")
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def resolve_dtype(dtype) -> torch.dtype:
    """Storage type of the device matrices.  Default (round 6): float64 - the reference's own arithmetic (its arrays are float64 from loompy on,
    analysis.py:59-61; correlations within 1e-10 of its kernels) - because a drop-in must first of all return what the reference returns.
    float32 (``dtype="float32"`` on a call or an object, ``VELOCYTO_AMD_DTYPE=float32`` for the process) is the opt-in production mode: half
    the memory, stage D 3 x faster, correlations within 5e-5."""
    if dtype is None:
        import os
        dtype = os.environ.get("VELOCYTO_AMD_DTYPE", "float64")
    if isinstance(dtype, torch.dtype):
        out = dtype
    else:
        out = {"float32": torch.float32, "f32": torch.float32, "float64": torch.float64, "f64": torch.float64,
               np.float32: torch.float32, np.float64: torch.float64}[dtype if not isinstance(dtype, np.dtype) else dtype.type]
    if out not in _DT:
        raise ValueError(f"unsupported dtype {dtype}")
    return out


def padded_ld(G: int) -> int:
    return (G + 63) // 64 * 64


class CellMatrix:
    """A (C cells, G genes) matrix on the device, rows padded to ``ld`` elements."""

    __slots__ = ("t", "G")

    def __init__(self, t: torch.Tensor, G: int):
        assert t.dim() == 2 and t.is_contiguous() and t.is_cuda and t.dtype in _DT and t.shape[1] >= G
        self.t, self.G = t, int(G)

    C = property(lambda self: int(self.t.shape[0]))
    ld = property(lambda self: int(self.t.shape[1]))
    dtype = property(lambda self: self.t.dtype)
    code = property(lambda self: _DT[self.t.dtype])

    @classmethod
    def empty(cls, C: int, G: int, dtype=None, zero_pad: bool = True) -> "CellMatrix":
        dev = require_gpu()
        ld = padded_ld(G)
        t = torch.empty((C, ld), dtype=resolve_dtype(dtype), device=dev)
        if zero_pad and ld > G:
            t[:, G:].zero_()
        return cls(t, G)

    @classmethod
    def from_genes_major(cls, a, dtype=None) -> "CellMatrix":
        """(G, C) numpy / torch array (any order, any float/int dtype) -> device cells-major."""
        dev = require_gpu()
        dt = resolve_dtype(dtype)
        if isinstance(a, CellMatrix):
            return a if a.dtype == dt else a.astype(dt)
        if isinstance(a, np.ndarray):
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            if a.flags.f_contiguous and not a.flags.c_contiguous:
                # Fortran (G,C) == C-order (C,G): already cells-major, no transpose needed
                src = torch.from_numpy(np.ascontiguousarray(a.T)).to(dev)
                return cls.from_cells_major(src, dt)
            src = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        else:
            src = a.to(dev)
            if src.dtype not in _DT:
                src = src.double()
            src = src.contiguous()
        G, C = src.shape
        out = cls(torch.empty((C, padded_ld(G)), dtype=dt, device=dev), G)
        _lib.check(_lib.lib().vcy_transpose(src.data_ptr(), out.t.data_ptr(), G, C, C, out.ld, _DT[src.dtype], out.code, _stream()), "transpose")
        return out

    @classmethod
    def from_cells_major(cls, a, dtype=None) -> "CellMatrix":
        """(C, G) array -> padded device matrix (a plain padded copy, no transpose)."""
        dev = require_gpu()
        dt = resolve_dtype(dtype)
        src = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
        src = src.to(dev)
        C, G = src.shape
        out = cls(torch.zeros((C, padded_ld(G)), dtype=dt, device=dev), G)
        out.t[:, :G] = src.to(dt)
        return out

    def astype(self, dtype) -> "CellMatrix":
        dt = resolve_dtype(dtype)
        return self if dt == self.dtype else CellMatrix(self.t.to(dt), self.G)

    def clone(self) -> "CellMatrix":
        return CellMatrix(self.t.clone(), self.G)

    def rows(self, c0: int, c1: int) -> "CellMatrix":
        return CellMatrix(self.t[c0:c1], self.G)

    def to_genes_major(self, dtype=np.float64, order: str = "C") -> np.ndarray:
        """Back to the reference's (G, C) numpy layout."""
        C, G = self.C, self.G
        if order == "F":   # Fortran (G,C) is exactly our memory order: plain copy
            return self.t[:, :G].to("cpu").numpy().astype(dtype, copy=False).T
        dst = torch.empty((G, C), dtype=self.dtype, device=self.t.device)
        _lib.check(_lib.lib().vcy_transpose(self.t.data_ptr(), dst.data_ptr(), C, G, self.ld, C, self.code, self.code, _stream()), "transpose")
        return dst.cpu().numpy().astype(dtype, copy=False)

    def to_cells_major(self, dtype=np.float64) -> np.ndarray:
        return self.t[:, : self.G].to("cpu").numpy().astype(dtype, copy=False)


class CountMatrix:
    """A (C cells, G genes) molecule-count matrix on the device (the loom layers as stored, velocyto/constants.py:11),
    cells-major, rows padded with zeros to a multiple of 64 elements.  uint16 lives in an int16 tensor (same bits; torch's
    uint16 support is partial); a layer whose counts all fit a byte may be held as uint8 (`narrowed()`, lossless) so
    that every gather of the pooling kernel moves half the bytes."""

    __slots__ = ("t", "G")

    def __init__(self, t: torch.Tensor, G: int):
        assert t.dim() == 2 and t.is_contiguous() and t.is_cuda and t.dtype in (torch.int16, torch.uint8) and t.shape[1] >= G and t.shape[1] % 16 == 0
        self.t, self.G = t, int(G)

    C = property(lambda self: int(self.t.shape[0]))
    ld = property(lambda self: int(self.t.shape[1]))
    code = property(lambda self: U8 if self.t.dtype == torch.uint8 else U16)

    @staticmethod
    def representable(a: np.ndarray) -> bool:
        a = np.asarray(a)
        if a.dtype == np.uint16 or a.dtype == np.uint8:
            return True
        if a.dtype.kind in "iu":
            return a.size == 0 or (a.min() >= 0 and a.max() <= 65535)
        return False

    @classmethod
    def from_genes_major(cls, a: np.ndarray, narrow: bool = True) -> "CountMatrix":
        """(G, C) integer counts (0..65535) -> device (C, ld) through the tiled transpose kernel; uint8 storage when every
        count fits a byte and `narrow`."""
        dev = require_gpu()
        a = np.asarray(a)
        if not cls.representable(a):
            raise ValueError("CountMatrix holds integer counts in 0..65535")
        G, C = a.shape
        if narrow and (a.size == 0 or int(a.max()) <= 255):
            src = torch.from_numpy(np.ascontiguousarray(a.astype(np.uint8, copy=False))).to(dev)
            out = torch.empty((C, padded_ld(G)), dtype=torch.uint8, device=dev)
            code = U8
        else:
            src = torch.from_numpy(np.ascontiguousarray(a.astype(np.uint16, copy=False)).view(np.int16)).to(dev)
            out = torch.empty((C, padded_ld(G)), dtype=torch.int16, device=dev)
            code = U16
        _lib.check(_lib.lib().vcy_transpose(src.data_ptr(), out.data_ptr(), G, C, C, out.shape[1], code, code, _stream()), "transpose(counts)")
        return cls(out, G)

    @classmethod
    def from_cells_major_tensor(cls, t: torch.Tensor, G: int) -> "CountMatrix":
        """(C, >=G) float/int device tensor holding integer counts -> padded uint16 copy."""
        C = t.shape[0]
        out = torch.zeros((C, padded_ld(G)), dtype=torch.int16, device=t.device)
        out[:, :G] = t[:, :G].to(torch.int32).to(torch.int16)       # wraps 32768..65535 onto the same 16 bits
        return cls(out, G)

    def as_int32(self, c0: int = 0, c1: Optional[int] = None) -> torch.Tensor:
        """Counts of rows c0:c1 as int32 (both storage widths)."""
        blk = self.t[c0:c1, : self.G]
        return blk.to(torch.int32) if blk.dtype == torch.uint8 else (blk.to(torch.int32) & 0xFFFF)

    def narrowed(self, block: int = 8192) -> "CountMatrix":
        """The same counts as uint8 if none exceeds 255 (else self)."""
        if self.t.dtype == torch.uint8:
            return self
        for s in range(0, self.C, block):
            if int((self.t[s:s + block].to(torch.int32) & 0xFFFF).max()) > 255:
                return self
        out = torch.empty(self.t.shape, dtype=torch.uint8, device=self.t.device)
        for s in range(0, self.C, block):
            out[s:s + block] = self.t[s:s + block].to(torch.uint8)      # values <= 255: the low byte is the count
        return CountMatrix(out, self.G)

    def to_float(self, dtype=None) -> CellMatrix:
        """float copy (counts as they are), cells-major."""
        dt = resolve_dtype(dtype)
        out = CellMatrix(torch.zeros((self.C, padded_ld(self.G)), dtype=dt, device=self.t.device), self.G)
        src = self.t if self.t.dtype == torch.uint8 else (self.t.to(torch.int32) & 0xFFFF)
        out.t[:, : self.ld] = src.to(dt)[:, : out.ld]
        return out


class CsrCounts:
    """A (C cells, G genes) molecule-count layer on the device in CSR form, cells-major: ``indptr`` (C + 1) int64,
    ``indices`` (nnz) int32 gene numbers ascending inside a row, ``data`` (nnz) counts as uint8 (no count above 255) or
    uint16 bits in an int16 tensor.  The atlas-scale storage of a loom layer (BASELINE.json configs[4]: 1M cells x 30k genes
    at ~8 % density is 2.4e9 non-zeros = 12-14 GB where the dense uint16 layer is 60 GB and the reference's float64 copy,
    analysis.py:59-61, 240 GB).  ``slabptr`` is the per-row table of gene-slab boundaries the pooling kernel walks
    (vcy_csr_slab_ptr), built on first use."""

    __slots__ = ("indptr", "indices", "data", "G", "_slabptr", "_nnz", "_istore", "_dstore")

    def __init__(self, indptr: torch.Tensor, indices: torch.Tensor, data: torch.Tensor, G: int):
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32 and data.dtype in (torch.uint8, torch.int16)
        assert indptr.is_cuda and indices.is_cuda and data.is_cuda and indices.numel() == data.numel()
        self._nnz = int(indices.numel())
        indices, data = indices.contiguous(), data.contiguous()
        self._istore, self._dstore = indices, data                # what the kernels are handed
        if indices.numel() < 4:                                   # vcy_knn_pool_csr reads quads of consecutive non-zeros: at least 4 stored
            pad = 4 - indices.numel()                             # elements (velocyto_hip.h); those past indptr[C] belong to no row
            self._istore = torch.cat([indices, torch.zeros(pad, dtype=indices.dtype, device=indices.device)])
            self._dstore = torch.cat([data, torch.zeros(pad, dtype=data.dtype, device=data.device)])
            indices, data = self._istore[: self._nnz], self._dstore[: self._nnz]
        self.indptr, self.indices, self.data, self.G = indptr.contiguous(), indices, data, int(G)
        self._slabptr = None

    C = property(lambda self: int(self.indptr.numel()) - 1)
    nnz = property(lambda self: self._nnz)
    code = property(lambda self: U8 if self.data.dtype == torch.uint8 else U16)
    nbytes = property(lambda self: self.indptr.numel() * 8 + self.indices.numel() * 4 + self.data.numel() * self.data.element_size())

    @property
    def slabptr(self) -> torch.Tensor:
        if self._slabptr is None:
            L = _lib.lib()
            nslab = (self.G + int(L.vcy_csr_slab_genes()) - 1) // int(L.vcy_csr_slab_genes())
            sp = torch.empty((max(self.C, 1), nslab + 1), dtype=torch.int32, device=self.indptr.device)
            if self.C:
                _lib.check(L.vcy_csr_slab_ptr(self.indptr.data_ptr(), self._istore.data_ptr(), sp.data_ptr(), self.C, self.G, _stream()), "csr_slab_ptr")
            self._slabptr = sp
        return self._slabptr

    @classmethod
    def from_dense(cls, m: "CountMatrix", block: int = 8192) -> "CsrCounts":
        """The non-zeros of a dense count matrix, row by row (block-wise: the index temporaries stay small)."""
        dev = m.t.device
        ptr, idx, dat = [torch.zeros(1, dtype=torch.int64, device=dev)], [], []
        base = 0
        for s0 in range(0, m.C, block):
            blk = m.t[s0:s0 + block, : m.G]
            nz = blk != 0
            cnt = nz.sum(1)
            ptr.append(base + torch.cumsum(cnt, 0))
            base += int(cnt.sum())
            rc = torch.nonzero(nz, as_tuple=False)                # row-major: rows ascending, genes ascending inside a row
            idx.append(rc[:, 1].to(torch.int32))
            dat.append(blk[nz])
        cat = lambda xs, dt: torch.cat(xs) if xs else torch.empty(0, dtype=dt, device=dev)
        return cls(torch.cat(ptr), cat(idx, torch.int32), cat(dat, m.t.dtype), m.G)

    @classmethod
    def from_scipy(cls, a, G: Optional[int] = None, narrow: bool = True) -> "CsrCounts":
        """scipy.sparse matrix of shape (C, G) holding integer counts 0..65535 -> device CSR (sorted indices, explicit zeros dropped)."""
        import scipy.sparse as sp
        dev = require_gpu()
        a = sp.csr_matrix(a)
        a.sum_duplicates()
        a.eliminate_zeros()
        a.sort_indices()
        vals = np.asarray(a.data)
        if vals.size and (vals.min() < 0 or vals.max() > 65535 or not np.array_equal(vals, np.rint(vals))):
            raise ValueError("CsrCounts holds integer counts in 0..65535")
        if narrow and (vals.size == 0 or vals.max() <= 255):
            dat = torch.from_numpy(vals.astype(np.uint8))
        else:
            dat = torch.from_numpy(vals.astype(np.uint16).view(np.int16))
        return cls(torch.from_numpy(a.indptr.astype(np.int64)).to(dev), torch.from_numpy(a.indices.astype(np.int32)).to(dev), dat.to(dev),
                   a.shape[1] if G is None else G)

    def rows(self, sel: torch.Tensor) -> "CsrCounts":
        """The CSR of the rows `sel` (int64 row numbers, any order, repeats allowed): the row gather behind cell blocks and
        count-row halos."""
        sel = sel.to(device=self.indptr.device, dtype=torch.int64)
        start, stop = self.indptr[sel], self.indptr[sel + 1]
        lens = stop - start
        ptr = torch.zeros(sel.numel() + 1, dtype=torch.int64, device=sel.device)
        torch.cumsum(lens, 0, out=ptr[1:])
        total = int(ptr[-1]) if sel.numel() else 0
        # position t of the output belongs to output row r = searchsorted(ptr, t, right) - 1 and source start[r] + (t - ptr[r])
        t = torch.arange(total, dtype=torch.int64, device=sel.device)
        r = torch.repeat_interleave(torch.arange(sel.numel(), device=sel.device), lens) if total else t
        src = start[r] + (t - ptr[r]) if total else t
        return CsrCounts(ptr, self.indices[src], self.data[src], self.G)

    def to_dense(self, narrow: bool = False) -> "CountMatrix":
        dev = self.indptr.device
        dt = torch.uint8 if (self.data.dtype == torch.uint8) else torch.int16
        out = torch.zeros((self.C, padded_ld(self.G)), dtype=dt, device=dev)
        if self.nnz:
            r = torch.repeat_interleave(torch.arange(self.C, device=dev), self.indptr[1:] - self.indptr[:-1])
            out[r, self.indices.long()] = self.data
        return CountMatrix(out, self.G)

    def row_sums(self) -> torch.Tensor:
        """Molecules per cell (S.sum(0) of the reference's (G, C) layout; analysis.py:540-546) as float64."""
        vals = self.data.to(torch.int64) if self.data.dtype == torch.uint8 else (self.data.to(torch.int64) & 0xFFFF)
        cs = torch.zeros(self.nnz + 1, dtype=torch.int64, device=self.indptr.device)
        torch.cumsum(vals, 0, out=cs[1:])
        return (cs[self.indptr[1:]] - cs[self.indptr[:-1]]).double()


def weight_rows_from_sorted_knn(idx_sorted: torch.Tensor, diag: float, dtype=None):
    """Device form of neighbors.weights_from_sorted_knn: the rows of knn_smoothing_w (analysis.py:1006-1010 on a kNN graph without
    zero distances) as (indptr int64, indices int32, values) - every row the cell itself (`diag`) and its k neighbours (1), in
    ascending cell number, times the reciprocal of the row sum k + diag: the same fp64 values scipy's chain produces."""
    dev = idx_sorted.device
    n, k = idx_sorted.shape
    me = torch.arange(n, device=dev, dtype=torch.int32)[:, None]
    cols = torch.cat([me, idx_sorted.to(torch.int32)], 1)
    vals = torch.ones((n, k + 1), dtype=torch.float64, device=dev)
    vals[:, 0] = float(diag)
    vals = vals * (1.0 / (np.float64(k) + np.float64(diag)))
    cols, vals = canonical_graph_rows(cols, vals)
    indptr = torch.arange(0, (n + 1) * (k + 1), k + 1, device=dev, dtype=torch.int64)
    return indptr, cols.reshape(-1).contiguous(), vals.reshape(-1).to(resolve_dtype(dtype)).contiguous()


def knn_pool_csr(counts: CsrCounts, scale, indptr, indices, weights, dtype=None, maximum: bool = False, cell0: int = 0,
                 C_out: Optional[int] = None, out: Optional[CellMatrix] = None, order: Optional[torch.Tensor] = None,
                 validate: bool = True) -> CellMatrix:
    """Pooled matrix gathered from a CSR count layer (vcy_knn_pool_csr): out[c,:] = sum_p w[p] * scale[idx[p]] * counts[idx[p],:],
    bit-identical to knn_pool_counts on the densified layer."""
    dev = counts.indptr.device
    dt = resolve_dtype(dtype)
    C_out = counts.C - cell0 if C_out is None else C_out
    ip = (indptr if isinstance(indptr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(indptr).astype(np.int64))).to(device=dev, dtype=torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    w = (weights if isinstance(weights, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(weights))).to(device=dev, dtype=dt).contiguous()
    sc = (torch.ones(counts.C, dtype=torch.float64, device=dev) if scale is None else
          (scale if isinstance(scale, torch.Tensor) else torch.as_tensor(np.asarray(scale, dtype=np.float64))).to(device=dev, dtype=torch.float64).contiguous())
    assert ip.numel() == C_out + 1 and ix.numel() == w.numel() and sc.numel() == counts.C
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= counts.C):
        raise ValueError("neighbour index out of range")
    out = CellMatrix.empty(C_out, counts.G, dt) if out is None else out
    assert out.C >= C_out and out.G == counts.G and out.dtype == dt
    if order is not None:
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        assert order.numel() == C_out
    _lib.check(_lib.lib().vcy_knn_pool_csr(counts.indptr.data_ptr(), counts._istore.data_ptr(), counts._dstore.data_ptr(), counts.slabptr.data_ptr(),
                                           sc.data_ptr(), out.t.data_ptr(), ip.data_ptr(), ix.data_ptr(), w.data_ptr(), _p(order), counts.C, counts.G,
                                           out.ld, cell0, C_out, int(maximum), counts.code, out.code, _stream()), "knn_pool_csr")
    return out


def localize_rows(ixs: torch.Tensor, b0: int, b1: int, outside: torch.Tensor) -> torch.Tensor:
    """Row numbers of a compact buffer laid out as [rows b0..b1-1 | the rows `outside` (ascending, disjoint from b0..b1-1)]
    for the global row numbers `ixs`; every index must be in one of the two parts."""
    g = ixs.long()
    own = (g >= b0) & (g < b1)
    if outside.numel():
        pos = torch.searchsorted(outside, g.reshape(-1)).reshape(g.shape)
        hit = outside[pos.clamp(max=outside.numel() - 1)] == g
        assert bool((own | hit).all()), "localize_rows: an index is in neither part of the buffer"
    else:
        pos = torch.zeros_like(g)
        assert bool(own.all()), "localize_rows: an index is outside the block and no outside rows were given"
    return torch.where(own, g - b0, (b1 - b0) + pos).to(torch.int32).contiguous()


def _as_i32(ixs, dev) -> torch.Tensor:
    if isinstance(ixs, torch.Tensor):
        return ixs.to(device=dev, dtype=torch.int32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(ixs).astype(np.int32)).to(dev)


# --------------------------------------------------------------------------- stage D
TILE_COLS = 256      # widest neighbour list one workgroup of the grouped kernel sorts in LDS (csrc/coldeltacor.hip)


def _sorted_rows(ix: torch.Tensor, out: torch.Tensor, presorted: Optional[bool] = None):
    """Lists wider than one tile are walked in column tiles; a pair's value does not depend on its column, so the rows are
    sorted by neighbour index first (adjacent cells then meet the same rows in the same tile) and the results are put
    back in the caller's column order.  Returns (ixs to launch with, (perm, sorted-order buffer) or None, caller's out).
    presorted: None = look (reads a flag back: a device sync); True / False = the caller knows (no sync)."""
    if ix.shape[1] <= TILE_COLS or (bool((ix[:, 1:] >= ix[:, :-1]).all()) if presorted is None else presorted):
        return ix, None, out
    srt, perm = torch.sort(ix, dim=1)
    # the launch buffer starts as the caller's rows in sorted column order: rows the schedule does not name round-trip
    return srt.contiguous(), (perm, out.gather(1, perm)), out


def abs_stats(e: CellMatrix) -> torch.Tensor:
    """Device tensor [sum |e|, smallest non-zero |e| (inf if none; denormals count), number of non-zero entries] over the WHOLE
    matrix (vcy_abs_stats: one streaming pass, ~1 ms per 6 GB)."""
    L = _lib.lib()
    out = torch.empty(3, dtype=torch.float64, device=e.t.device)
    ws = torch.empty(int(L.vcy_abs_stats_workspace_bytes()), dtype=torch.uint8, device=e.t.device)
    _lib.check(L.vcy_abs_stats(e.t.data_ptr(), out.data_ptr(), ws.data_ptr(), e.C, e.G, e.ld, e.code, _stream()), "abs_stats")
    return out


F64_SQRT_MAX = 1e38


def check_f64_sqrt_domain(e: CellMatrix) -> None:
    """The f64 partial-sqrt element (csrc/coldeltacor.hip, xform<double, SQRT, PARTIAL>) seeds its square root from the f32 unit: the
    correctly rounded root in all but near-tie cases for arguments inside the f32 exponent range, NaN above 3.4e38 (the converted
    argument is +inf, v_rsq_f32 of it is 0, and inf * 0 is NaN).  A count-derived matrix never comes near it; a matrix that does is
    refused here (one reduction, one device->host sync - the callers that decide the branch rule once per matrix come through here,
    ops.partial_rules_for; every `validate=True` call of the partial entry points, fused ones included, does too)."""
    lo, hi = e.t.aminmax()
    m = max(abs(float(lo)), abs(float(hi)))
    if m >= F64_SQRT_MAX:
        raise ValueError(f"colDeltaCor sqrt transform in f64: |e| reaches {m:.3g}, outside the supported range (< {F64_SQRT_MAX:g})")


def literal_rule_forced() -> bool:
    """VELOCYTO_AMD_LITERAL_RULE=1: never pick the no-pseudocount form (the facade's `literal_rule` attribute does the same per object)."""
    import os
    return os.environ.get("VELOCYTO_AMD_LITERAL_RULE", "0") == "1"


def partial_rules_for(e: CellMatrix, transform: int, psc: float, stats: Optional[torch.Tensor] = None, cells: Optional[int] = None,
                      literal: bool = False, domain_checked: bool = False) -> int:
    """The rules value the callers of the *partial kernels pass for the reference's partial rule on this matrix:
    RULES_PARTIAL_NOPSC (A = sign(t) sqrt|t|, three instructions per gene instead of five) for the sqrt transform on an f32
    matrix when the pseudocount cannot be told from zero at the matrix's scale, else RULES_PARTIAL, the literal rule.

    The decision is a FACT about the whole matrix, not a sample: vcy_abs_stats reduces every entry (one pass, one device->host
    sync - callers decide once per matrix, not per launch).  NOPSC needs all of
      * psc <= 1e-9,
      * mean |e| >= 1e-4 (below that the pseudocount is a visible part of |t| + psc),
      * no non-zero |e| below 1e-20: v_rsq_f32 reads a denormal as zero, so t * rsq|t| of a difference below 2^-126 would be
        inf; differences of entries at or above 1e-20 are multiples of their ulp (> 1e-27) and stay in the normal range.
    `stats`: a precomputed abs_stats vector - sharded callers all-reduce it (sum, min, sum) so that every rank decides alike
    (distributed.all_reduce_abs_stats); `cells`: the number of cells the sums cover when it is not e.C.  `literal=True` or
    VELOCYTO_AMD_LITERAL_RULE=1 keeps the literal rule whatever the data.

    A-priori bound.  The two rules differ per gene by delta_g = sqrt(|t_g| + psc) - sqrt|t_g| <= min(sqrt(psc), psc / (2 sqrt|t_g|)),
    and not at all in f32 once |t_g| >= 2^24 psc (`|t| + psc` rounds to `|t|`; 1.7e-3 at the default 1e-10).  Pearson's r of the
    pair moves by |dr| <= 2 ||delta||_2 / ||A - mean A||_2 to first order (Cauchy-Schwarz on the centred, normalised vectors),
    i.e. <= psc * sqrt(mean_g 1/|t_g|) / sd(A) over the genes with 0 < |t_g| < 2^24 psc: a few 1e-7 on count-scale data
    (measured 1.5e-7 over all 12.5 M pairs of the bench workload), and below the f32 tolerance of 1e-5 down to matrix scales of
    1e-4 (tests/test_gpu_ops.py::test_partial_nopsc_rule_bound_on_scaled_matrices)."""
    if transform == SQRT and e.dtype == torch.float64 and e.C and not domain_checked:     # (domain_checked: the caller ran check_f64_sqrt_domain on
        check_f64_sqrt_domain(e)                                                            #  the matrix itself - `e` may be a staging buffer, atlas.py)
    if transform != SQRT or e.dtype != torch.float32 or not (0.0 <= float(psc) <= PSC_NEGLIGIBLE) or e.C == 0 or literal or literal_rule_forced():
        return RULES_PARTIAL
    st = abs_stats(e) if stats is None else stats
    total, tiny, _ = (float(x) for x in st.cpu())
    mean = total / (float(e.C if cells is None else cells) * e.G)
    return RULES_PARTIAL_NOPSC if (mean >= SCALE_ORDINARY and tiny >= 1e-20) else RULES_PARTIAL


def coldeltacor_partial(e: CellMatrix, d: CellMatrix, ixs, transform: int, rules: int = RULES_PARTIAL, psc: float = 0.0,
                        cell0: int = 0, order: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                        d_row0: int = 0, validate: bool = True, presorted: Optional[bool] = None) -> torch.Tensor:
    """Compact correlations out[c, n] = corr(cell0 + c, ixs[c, n]); ixs: (C_out, nrndm).
    `d` may hold only the rows of cells d_row0.. (cell-sharded runs); `validate` range-checks ixs
    (a device->host sync; hot loops that built ixs themselves pass False)."""
    assert e.ld == d.ld and e.dtype == d.dtype and e.G == d.G
    ix = _as_i32(ixs, e.t.device)
    C_out, nrndm = ix.shape
    assert d_row0 <= cell0 and cell0 + C_out <= d_row0 + d.C
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= e.C):
        raise ValueError("neighbour index out of range")
    if validate and transform == SQRT and e.dtype == torch.float64 and e.C:
        check_f64_sqrt_domain(e)
    if out is None:
        out = torch.empty((C_out, nrndm), dtype=e.dtype, device=e.t.device)
    n_sched = C_out
    if order is not None:
        order = order.to(device=e.t.device, dtype=torch.int32).contiguous()
        n_sched = int(order.numel())          # a schedule over a subset of the cells: the other rows of `out` stay as they are
        assert n_sched <= C_out
        if n_sched == 0:
            return out
    ix, perm, user_out = _sorted_rows(ix, out, presorted)
    _lib.check(_lib.lib().vcy_coldeltacor_partial(e.t.data_ptr(), d.t.data_ptr(), ix.data_ptr(), (out if perm is None else perm[1]).data_ptr(), _p(order),
                                                  e.C, e.G, e.ld, cell0, n_sched, d_row0, nrndm, transform, rules, float(psc),
                                                  e.code, _stream()), "coldeltacor_partial")
    return out if perm is None else user_out.scatter_(1, perm[0], perm[1])


def coldeltacor_partial_fused(Sx: CellMatrix, Ux: CellMatrix, gamma: torch.Tensor, q: Optional[torch.Tensor], ixs, transform: int,
                              rules: int = RULES_PARTIAL, psc: float = 0.0, dt_shift: float = 1.0, used_dt: float = 1.0, cell0: int = 0,
                              u_row0: int = 0, order: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                              validate: bool = True) -> torch.Tensor:
    """Stage C + D in one launch: correlations against the dmat the velocity chain would produce (vcy_coldeltacor_partial_fused)."""
    assert Sx.ld == Ux.ld and Sx.dtype == Ux.dtype and Sx.G == Ux.G
    dev = Sx.t.device
    ix = _as_i32(ixs, dev)
    C_out, nrndm = ix.shape
    assert u_row0 <= cell0 and cell0 + C_out <= u_row0 + Ux.C
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= Sx.C):
        raise ValueError("neighbour index out of range")
    if validate and transform == SQRT and Sx.dtype == torch.float64 and Sx.C:
        check_f64_sqrt_domain(Sx)
    if out is None:
        out = torch.empty((C_out, nrndm), dtype=Sx.dtype, device=dev)
    gamma = gamma.to(device=dev, dtype=torch.float32).contiguous()
    q = None if q is None else q.to(device=dev, dtype=torch.float32).contiguous()
    n_sched = C_out
    if order is not None:
        # the schedule may name only SOME of the C_out cells (e.g. the cells whose neighbours are all rank-local, while the
        # halo exchange is still in flight): rows of ixs / out are addressed by cell id, the others are left untouched
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        n_sched = int(order.numel())
        assert n_sched <= C_out
        if n_sched == 0:
            return out
    ix, perm, user_out = _sorted_rows(ix, out)
    _lib.check(_lib.lib().vcy_coldeltacor_partial_fused(Sx.t.data_ptr(), Ux.t.data_ptr(), gamma.data_ptr(), _p(q), ix.data_ptr(),
                                                        (out if perm is None else perm[1]).data_ptr(),
                                                        _p(order), Sx.C, Sx.G, Sx.ld, cell0, n_sched, u_row0, nrndm, transform, rules, float(psc),
                                                        float(dt_shift), float(used_dt), Sx.code, _stream()), "coldeltacor_partial_fused")
    return out if perm is None else user_out.scatter_(1, perm[0], perm[1])


def _sched(order, dev, C_out):
    """(order tensor or None, number of scheduled cells)."""
    if order is None:
        return None, C_out
    order = order.to(device=dev, dtype=torch.int32).contiguous()
    assert int(order.numel()) <= C_out
    return order, int(order.numel())


def coldeltacor_partial_dual(e: CellMatrix, d: CellMatrix, d_rndm: CellMatrix, ixs, transform: int, rules: int = RULES_PARTIAL,
                             psc: float = 0.0, cell0: int = 0, order: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                             out_rndm: Optional[torch.Tensor] = None, d_row0: int = 0, validate: bool = True,
                             presorted: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(corr, corr_rndm): the real and the randomised-control correlations of a neighbour list in one pass
    (vcy_coldeltacor_partial_dual; analysis.py:1539-1542, 1578-1601)."""
    assert e.ld == d.ld == d_rndm.ld and e.dtype == d.dtype == d_rndm.dtype and e.G == d.G == d_rndm.G and d.C == d_rndm.C
    ix = _as_i32(ixs, e.t.device)
    C_out, nrndm = ix.shape
    assert d_row0 <= cell0 and cell0 + C_out <= d_row0 + d.C
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= e.C):
        raise ValueError("neighbour index out of range")
    if validate and transform == SQRT and e.dtype == torch.float64 and e.C:
        check_f64_sqrt_domain(e)
    out = torch.empty((C_out, nrndm), dtype=e.dtype, device=e.t.device) if out is None else out
    out_rndm = torch.empty((C_out, nrndm), dtype=e.dtype, device=e.t.device) if out_rndm is None else out_rndm
    order, n_sched = _sched(order, e.t.device, C_out)
    if n_sched == 0:
        return out, out_rndm
    ix, perm, _ = _sorted_rows(ix, out, presorted)
    perm2 = None if perm is None else out_rndm.gather(1, perm[0])
    _lib.check(_lib.lib().vcy_coldeltacor_partial_dual(e.t.data_ptr(), d.t.data_ptr(), d_rndm.t.data_ptr(), ix.data_ptr(),
                                                       (out if perm is None else perm[1]).data_ptr(), (out_rndm if perm is None else perm2).data_ptr(),
                                                       _p(order), e.C, e.G, e.ld, cell0, n_sched, d_row0, nrndm, transform, rules, float(psc),
                                                       e.code, _stream()), "coldeltacor_partial_dual")
    if perm is not None:
        out.scatter_(1, perm[0], perm[1])
        out_rndm.scatter_(1, perm[0], perm2)
    return out, out_rndm


def coldeltacor_partial_fused_dual(Sx: CellMatrix, Ux: CellMatrix, gamma: torch.Tensor, q: Optional[torch.Tensor], d_rndm: CellMatrix, ixs,
                                   transform: int, rules: int = RULES_PARTIAL, psc: float = 0.0, dt_shift: float = 1.0, used_dt: float = 1.0,
                                   cell0: int = 0, u_row0: int = 0, order: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                                   out_rndm: Optional[torch.Tensor] = None, validate: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stage C + D + the randomised control in one launch (vcy_coldeltacor_partial_fused_dual)."""
    assert Sx.ld == Ux.ld == d_rndm.ld and Sx.dtype == Ux.dtype == d_rndm.dtype and Sx.G == Ux.G == d_rndm.G and Ux.C == d_rndm.C
    dev = Sx.t.device
    ix = _as_i32(ixs, dev)
    C_out, nrndm = ix.shape
    assert u_row0 <= cell0 and cell0 + C_out <= u_row0 + Ux.C
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= Sx.C):
        raise ValueError("neighbour index out of range")
    if validate and transform == SQRT and Sx.dtype == torch.float64 and Sx.C:
        check_f64_sqrt_domain(Sx)
    out = torch.empty((C_out, nrndm), dtype=Sx.dtype, device=dev) if out is None else out
    out_rndm = torch.empty((C_out, nrndm), dtype=Sx.dtype, device=dev) if out_rndm is None else out_rndm
    gamma = gamma.to(device=dev, dtype=torch.float32).contiguous()
    q = None if q is None else q.to(device=dev, dtype=torch.float32).contiguous()
    order, n_sched = _sched(order, dev, C_out)
    if n_sched == 0:
        return out, out_rndm
    ix, perm, _ = _sorted_rows(ix, out)
    perm2 = None if perm is None else out_rndm.gather(1, perm[0])
    _lib.check(_lib.lib().vcy_coldeltacor_partial_fused_dual(Sx.t.data_ptr(), Ux.t.data_ptr(), gamma.data_ptr(), _p(q), d_rndm.t.data_ptr(), ix.data_ptr(),
                                                             (out if perm is None else perm[1]).data_ptr(), (out_rndm if perm is None else perm2).data_ptr(),
                                                             _p(order), Sx.C, Sx.G, Sx.ld, cell0, n_sched, u_row0, nrndm, transform, rules, float(psc),
                                                             float(dt_shift), float(used_dt), Sx.code, _stream()), "coldeltacor_partial_fused_dual")
    if perm is not None:
        out.scatter_(1, perm[0], perm[1])
        out_rndm.scatter_(1, perm[0], perm2)
    return out, out_rndm


def coldeltacor_full(e: CellMatrix, d: CellMatrix, transform: int, psc: float = 0.0, cell0: int = 0,
                     C_out: Optional[int] = None, rm: Optional[torch.Tensor] = None, accumulate: bool = False, validate: bool = True) -> torch.Tensor:
    """Dense correlation rows rm[c, i] for c in [cell0, cell0+C_out), all i (speedboosted._colDeltaCor / Sqrt / Log10,
    speedboosted.pyx:13-257).  `validate` (linear variant on the matrix cores only): check that the padding columns G .. ld - 1 of
    e and d are zero, which that kernel's contraction over whole 16-gene slabs relies on (one small device reduction + a sync)."""
    assert e.t.shape == d.t.shape and e.dtype == d.dtype
    C_out = e.C - cell0 if C_out is None else C_out
    if rm is None:
        rm = torch.zeros((C_out, e.C), dtype=e.dtype, device=e.t.device)
        accumulate = False
    assert rm.is_contiguous() and rm.shape == (C_out, e.C) and rm.dtype == e.dtype
    L = _lib.lib()
    # the matrix-core kernel walks whole 16-gene slabs: a row pitch that does not hold them goes to the element-wise kernel
    if transform == LINEAR and FULL_LINEAR_MFMA and e.ld % 16 == 0 and e.t.data_ptr() % 16 == 0 and d.t.data_ptr() % 16 == 0:
        if validate and e.ld > e.G and bool(((e.t[:, e.G:] != 0) | (d.t[:, e.G:] != 0)).any()):     # (NaN != 0 as well)
            raise ValueError("coldeltacor_full: the padding columns G .. ld - 1 of e and d must be zero (CellMatrix.empty(zero_pad=True), "
                             "from_genes_major and every kernel of this library write them so)")
        # the one dense block contraction of the path (E E^T and D E^T over the genes): f64 matrix cores, Pearson epilogue fused,
        # nearly identical cells re-evaluated in the reference's centred form by the entry's own repair launch
        ws = torch.empty(int(L.vcy_coldeltacor_full_linear_workspace_bytes(e.C, C_out)), dtype=torch.uint8, device=e.t.device)
        _lib.check(L.vcy_coldeltacor_full_linear(e.t.data_ptr(), d.t.data_ptr(), rm.data_ptr(), ws.data_ptr(), e.C, e.G, e.ld, cell0, C_out,
                                                 rm.shape[1], int(accumulate), e.code, _stream()), "coldeltacor_full_linear")
        return rm
    _lib.check(L.vcy_coldeltacor_full(e.t.data_ptr(), d.t.data_ptr(), rm.data_ptr(), e.C, e.G, e.ld, cell0, C_out,
                                      rm.shape[1], transform, float(psc), int(accumulate), e.code, _stream()),
               "coldeltacor_full")
    return rm


FULL_LINEAR_MFMA = True    # the all-pairs LINEAR variant on the f64 matrix cores (vcy_coldeltacor_full_linear); False = the element-wise kernel


def scatter_rows(vals: torch.Tensor, ixs, ncols: int, rm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rm[c, ixs[c,n]] += vals[c,n]  (the reference's dense (C,C) `rm`, speedboosted.pyx:332-336)."""
    ix = _as_i32(ixs, vals.device)
    C_out, nrndm = ix.shape
    if rm is None:
        rm = torch.zeros((C_out, ncols), dtype=vals.dtype, device=vals.device)
    _lib.check(_lib.lib().vcy_scatter_rows(vals.contiguous().data_ptr(), ix.data_ptr(), rm.data_ptr(), C_out, nrndm, rm.shape[1],
                                           _DT[vals.dtype], _stream()), "scatter_rows")
    return rm


# --------------------------------------------------------------------------- stage A
def canonical_graph_rows(indices: torch.Tensor, weights: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fixed-width graph rows (C, k + 1) with every row sorted by cell number (weights permuted alike).  The pooling kernels add
    a row's terms in list order, so the pooled vector depends on that order in its last bits.  The reference pools through
    scipy with column-sorted rows (`(knn > 0)` / `setdiag`, analysis.py:1006-1008, sort the CSR in place): two cells with the
    SAME closed neighbourhood - mutual neighbours inside a tight cluster - get bitwise equal vectors there, their difference is
    exactly zero on every gene and the zero rule of speedboosted.pyx:372 turns the pair into the NaN the facade then maps to 1
    (analysis.py:1605-1606).  Lists kept nearest-first instead give such a pair rounding noise of 1e-16 relative, which the
    partial-sqrt transform lifts to +-sqrt(psc) on every gene: a finite, meaningless correlation.  Device-built graphs (bench.py,
    atlas.py) therefore pool in this canonical order; the facade hands scipy's own (sorted) CSR through."""
    srt, perm = torch.sort(indices, dim=1)
    return srt.contiguous(), weights.gather(1, perm).contiguous()


def knn_pool(data: CellMatrix, indptr, indices, weights, maximum: bool = False, cell0: int = 0,
             C_out: Optional[int] = None, slab_genes: int = 0, out: Optional[CellMatrix] = None,
             validate: bool = True, order: Optional[torch.Tensor] = None) -> CellMatrix:
    """out[c,:] = sum_p w[p] data[indices[p],:] over CSR row c (neighbors.py:416-423 on device)."""
    dev = data.t.device
    C_out = data.C - cell0 if C_out is None else C_out
    ip = (indptr if isinstance(indptr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(indptr).astype(np.int64))).to(device=dev, dtype=torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    w = (weights if isinstance(weights, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(weights))).to(device=dev, dtype=data.dtype).contiguous()
    assert ip.numel() == C_out + 1 and ix.numel() == w.numel()
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= data.C):
        raise ValueError("neighbour index out of range")
    if out is None:
        out = CellMatrix.empty(C_out, data.G, data.dtype)
    if order is not None:
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        assert order.numel() == C_out
    _lib.check(_lib.lib().vcy_knn_pool(data.t.data_ptr(), out.t.data_ptr(), ip.data_ptr(), ix.data_ptr(), w.data_ptr(), _p(order), data.C,
                                       data.G, data.ld, cell0, C_out, int(maximum), int(slab_genes), data.code, _stream()), "knn_pool")
    return out


def knn_pool_w2(data: CellMatrix, indptr, indices, weights, weights2, cell0: int = 0, C_out: Optional[int] = None, slab_genes: int = 0,
                validate: bool = True, order: Optional[torch.Tensor] = None) -> Tuple[CellMatrix, CellMatrix]:
    """One matrix pooled with two weight sets over the same graph (vcy_knn_pool_w2: the rows are gathered once):
    out[c,:] = sum_p w[p] data[indices[p],:],  out2[c,:] = sum_p w2[p] data[indices[p],:]."""
    dev = data.t.device
    C_out = data.C - cell0 if C_out is None else C_out
    ip = (indptr if isinstance(indptr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(indptr).astype(np.int64))).to(device=dev, dtype=torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    as_w = lambda t: (t if isinstance(t, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(t))).to(device=dev, dtype=data.dtype).contiguous()
    w, w2 = as_w(weights), as_w(weights2)
    assert ip.numel() == C_out + 1 and ix.numel() == w.numel() == w2.numel()
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= data.C):
        raise ValueError("neighbour index out of range")
    out, out2 = CellMatrix.empty(C_out, data.G, data.dtype), CellMatrix.empty(C_out, data.G, data.dtype)
    if order is not None:
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        assert order.numel() == C_out
    _lib.check(_lib.lib().vcy_knn_pool_w2(data.t.data_ptr(), out.t.data_ptr(), out2.t.data_ptr(), ip.data_ptr(), ix.data_ptr(), w.data_ptr(),
                                          w2.data_ptr(), _p(order), data.C, data.G, data.ld, cell0, C_out, int(slab_genes), data.code, _stream()),
               "knn_pool_w2")
    return out, out2


def knn_pool2(data: CellMatrix, data2: CellMatrix, indptr, indices, weights, maximum: bool = False, cell0: int = 0,
              C_out: Optional[int] = None, slab_genes: int = 0, out: Optional[CellMatrix] = None, out2: Optional[CellMatrix] = None,
              validate: bool = True, order: Optional[torch.Tensor] = None) -> Tuple[CellMatrix, CellMatrix]:
    """Pool two matrices that share the weights (S and U of knn_imputation) in one launch."""
    dev = data.t.device
    assert data.t.shape == data2.t.shape and data.dtype == data2.dtype and data.G == data2.G
    C_out = data.C - cell0 if C_out is None else C_out
    ip = (indptr if isinstance(indptr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(indptr).astype(np.int64))).to(device=dev, dtype=torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    w = (weights if isinstance(weights, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(weights))).to(device=dev, dtype=data.dtype).contiguous()
    assert ip.numel() == C_out + 1 and ix.numel() == w.numel()
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= data.C):
        raise ValueError("neighbour index out of range")
    out = CellMatrix.empty(C_out, data.G, data.dtype) if out is None else out
    out2 = CellMatrix.empty(C_out, data.G, data.dtype) if out2 is None else out2
    if order is not None:
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        assert order.numel() == C_out
    _lib.check(_lib.lib().vcy_knn_pool2(data.t.data_ptr(), out.t.data_ptr(), data2.t.data_ptr(), out2.t.data_ptr(), ip.data_ptr(), ix.data_ptr(),
                                        w.data_ptr(), _p(order), data.C, data.G, data.ld, cell0, C_out, int(maximum), int(slab_genes), data.code,
                                        _stream()), "knn_pool2")
    return out, out2


def knn_pool_counts(cS: CountMatrix, cU: Optional[CountMatrix], scaleS, scaleU, indptr, indices, weights, dtype=None,
                    maximum: bool = False, cell0: int = 0, C_out: Optional[int] = None, slab_genes: int = 0,
                    out: Optional[CellMatrix] = None, out2: Optional[CellMatrix] = None, order: Optional[torch.Tensor] = None,
                    validate: bool = True):
    """Pooled Sx (and Ux) gathered straight from the uint16 count matrices with per-cell size factors
    (vcy_knn_pool_counts): out[c,:] = sum_p w[p] * scale[idx[p]] * counts[idx[p],:]."""
    dev = cS.t.device
    dt = resolve_dtype(dtype)
    C_out = cS.C - cell0 if C_out is None else C_out
    ip = (indptr if isinstance(indptr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(indptr).astype(np.int64))).to(device=dev, dtype=torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    w = (weights if isinstance(weights, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(weights))).to(device=dev, dtype=dt).contiguous()
    f64 = lambda t: None if t is None else (t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t, dtype=np.float64))).to(device=dev, dtype=torch.float64).contiguous()
    sS = f64(scaleS) if scaleS is not None else torch.ones(cS.C, dtype=torch.float64, device=dev)
    sU = None
    if cU is not None:
        assert cU.G == cS.G
        if cU.t.dtype != cS.t.dtype:          # one launch pools both layers: bring them to the wider storage
            widen = lambda m: m if m.t.dtype == torch.int16 else CountMatrix(m.t.to(torch.int16), m.G)
            cS, cU = widen(cS), widen(cU)
        assert cU.t.shape == cS.t.shape
        sU = f64(scaleU) if scaleU is not None else torch.ones(cS.C, dtype=torch.float64, device=dev)
    assert ip.numel() == C_out + 1 and ix.numel() == w.numel() and sS.numel() == cS.C
    if validate and ix.numel() and (int(ix.min()) < 0 or int(ix.max()) >= cS.C):
        raise ValueError("neighbour index out of range")
    out = CellMatrix.empty(C_out, cS.G, dt) if out is None else out
    if cU is not None:
        out2 = CellMatrix.empty(C_out, cS.G, dt) if out2 is None else out2
    if order is not None:
        order = order.to(device=dev, dtype=torch.int32).contiguous()
        assert order.numel() == C_out
    _lib.check(_lib.lib().vcy_knn_pool_counts(cS.t.data_ptr(), None if cU is None else cU.t.data_ptr(), sS.data_ptr(), _p(sU), out.t.data_ptr(),
                                              None if cU is None else out2.t.data_ptr(), ip.data_ptr(), ix.data_ptr(), w.data_ptr(), _p(order),
                                              cS.C, cS.G, cS.ld, out.ld, cell0, C_out, int(maximum), int(slab_genes), cS.code, out.code, _stream()),
               "knn_pool_counts")
    return (out, out2) if cU is not None else out


def _knn_query_block(L, C: int, Q: int, k: int, query_block: int, budget: int = 8 << 30) -> int:
    """Queries per launch: everything at once (up to `budget` bytes of scratch rows) when the search keeps its candidates
    in registers, `query_block` when it materialises the distance rows."""
    if not L.vcy_knn_row_free(C, k):
        return min(Q, query_block)
    return min(Q, max(query_block, (2 * budget // max(C, 1)) // 8 * 8))


def knn_search(space, k: int, include_self: bool = False, q0: int = 0, Q: Optional[int] = None,
               query_block: int = 8192) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact Euclidean kNN of rows q0..q0+Q of `space` (C, P) among all C rows.
    Returns (idx int32 (Q,k), dist float64 (Q,k)), nearest first, ties by index."""
    dev = require_gpu()
    x64 = (torch.from_numpy(np.array(space, dtype=np.float64, order="C")) if not isinstance(space, torch.Tensor) else space.double()).to(dev).contiguous()
    C, P = x64.shape
    Q = C - q0 if Q is None else Q
    if C > KNN_SEGMENT and k + 9 <= 128:
        return _knn_search_segmented(x64, k, include_self, q0, Q, query_block)
    ldx = (C + 63) // 64 * 64
    xt = torch.zeros((P, ldx), dtype=torch.float32, device=dev)
    xt[:, :C] = x64.t().float()
    idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    dist = torch.empty((Q, k), dtype=torch.float64, device=dev)
    L = _lib.lib()
    # the register-resident search needs one scratch row per 8 queries: all queries in ONE launch (6250 workgroups at 50k
    # cells, so that the selection phase of one workgroup overlaps the distance phase of others); the row-materialising
    # search holds (queries x C) distances and walks the queries in blocks
    qb = _knn_query_block(L, C, Q, k, query_block)
    ws = torch.empty(int(L.vcy_knn_workspace_bytes(C, qb, k)), dtype=torch.uint8, device=dev)
    for s in range(0, Q, qb):
        n = min(qb, Q - s)
        _lib.check(L.vcy_knn_search(xt.data_ptr(), x64.data_ptr(), idx[s:s + n].data_ptr(), dist[s:s + n].data_ptr(), ws.data_ptr(),
                                    C, P, ldx, q0 + s, n, k, int(include_self), _stream()), "knn_search")
    return idx, dist


# The row-free search kernel keeps a candidate's position inside its thread's strided slice in the low 10 mantissa bits of
# the tracked distance words, which caps one launch at 1022 x 256 candidates; larger point sets (atlas scale) are searched
# segment by segment and the per-segment lists merged.  The alternative inside the library - materialising the (queries x
# candidates) distance rows - measured 9 x slower per candidate at 400k cells.
KNN_SEGMENT = 1008 * 256


def _knn_search_segmented(x64: torch.Tensor, k: int, include_self: bool, q0: int, Q: int, query_block: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact kNN over more than KNN_SEGMENT candidates: the queries against every segment of the point set (vcy_knn_query,
    k + 1 nearest, nothing excluded), then a merge by (distance, index) and the removal of the query itself.  Distances are
    the same exact fp64 sums the one-launch search returns, every segment's list is sorted by (distance, index) and segments
    are visited in index order, so a stable sort by distance reproduces the one-launch order, ties by index included."""
    C = x64.shape[0]
    kk = k if include_self else k + 1
    qs = x64[q0:q0 + Q]
    ds, ix = [], []
    # segment bounds: a last segment too short to hold kk candidates is evened out with the one before it (no overlap, so no
    # candidate is ever listed twice - an overlap would need a dedup that exact distance ties can defeat)
    bounds = list(range(0, C, KNN_SEGMENT)) + [C]
    if len(bounds) > 2 and bounds[-1] - bounds[-2] <= kk:
        bounds[-2] = (bounds[-3] + bounds[-1]) // 2
    for s0, s1 in zip(bounds[:-1], bounds[1:]):
        i, d = knn_query(x64[s0:s1], qs, kk, query_block)
        ds.append(d)
        ix.append(i + s0)
    d_all, order = torch.sort(torch.cat(ds, 1), dim=1, stable=True)
    i_all = torch.gather(torch.cat(ix, 1), 1, order)
    keep = torch.ones_like(i_all, dtype=torch.bool)
    if not include_self:
        keep = i_all != torch.arange(q0, q0 + Q, device=i_all.device, dtype=i_all.dtype)[:, None]
    pos = torch.cumsum(keep.to(torch.int32), 1)
    sel = keep & (pos <= k)
    rows, cols = torch.nonzero(sel, as_tuple=True)
    idx = torch.empty((Q, k), dtype=torch.int32, device=x64.device)
    dist = torch.empty((Q, k), dtype=torch.float64, device=x64.device)
    dst = (pos[rows, cols] - 1).long()
    idx[rows, dst] = i_all[rows, cols]
    dist[rows, dst] = d_all[rows, cols]
    return idx, dist


PRUNED_STREAMS = 4          # concurrent tile searches of knn_search_pruned
PRUNED_SEGMENT = KNN_SEGMENT  # candidates one tile search takes in one launch (the row-free kernel's limit); more are searched in pieces


def knn_search_pruned(space, k: int, q0: int = 0, Q: Optional[int] = None, tile: int = 4096, stats: Optional[dict] = None
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact kNN (query excluded) with projection pruning for large point sets: the same result as knn_search, from far fewer
    distance evaluations when the two leading coordinates separate the points (PCA scores: they carry the most variance).

    ||q - x|| >= ||q[:2] - x[:2]||, so a point whose projection lies farther than R from a query cannot be among its k
    nearest once k points within R are known.  Queries are cut into tiles of `tile` points that are compact in the projection
    (sort by the first coordinate into strips, by the second inside a strip); a tile is searched (vcy_knn_query, brute force,
    exact fp64 distances) against the points of the grid cells that its bounding box grown by R touches, and the search is
    accepted only if every query's k-th distance is <= R - otherwise that tile is searched again with R = 1.05 x the largest
    k-th distance it found (which then holds by construction).  R starts from the k-th distances of a pilot sample searched
    against everything.  Candidates keep their global order inside a tile's subset, so ties come out by global index as in
    knn_search.  The host synchronises a fixed number of times (pilot, grid, boxes, one verdict per pass), not once per tile."""
    dev = require_gpu()
    x64 = (torch.from_numpy(np.array(space, dtype=np.float64, order="C")) if not isinstance(space, torch.Tensor) else space.double()).to(dev).contiguous()
    C, P = x64.shape
    Q = C - q0 if Q is None else Q
    assert 0 < k < C and P >= 2
    z = x64[:, :2].contiguous()
    # ---- pilot: k-th distance of a spread sample -> a starting radius
    npilot = min(Q, 2048)
    pilot = q0 + (torch.arange(npilot, device=dev) * (Q / npilot)).long()
    _, dp = knn_query(x64, x64[pilot], k + 1)
    zmin, zmax = z.min(0).values, z.max(0).values
    R0, x0, y0, x1, y1 = (float(v) for v in torch.stack([dp[:, k].max() * 1.1, zmin[0], zmin[1], zmax[0], zmax[1]]).cpu())
    # a tile holds thousands of queries: the radius starts from the pilot's LARGEST k-th distance
    # ---- uniform grid over the projection, cell ~ R0 / 8 (at most 1024 x 1024 cells); points sorted by cell, ascending global
    #      number inside a cell (stable sort), so that the candidates of a box are a few contiguous slices - no per-tile compaction
    #      (coarse cells cost evaluations: a box grown by R covers ~2 R per side, a cell of R would add 50 % of candidates)
    h = max(R0 / 8.0, (x1 - x0) / 1024.0, (y1 - y0) / 1024.0, 1e-300)
    ncx, ncy = int((x1 - x0) / h) + 1, int((y1 - y0) / h) + 1
    cx = ((z[:, 0] - x0) / h).long().clamp_(0, ncx - 1)
    cy = ((z[:, 1] - y0) / h).long().clamp_(0, ncy - 1)
    cell = cy * ncx + cx
    by_cell = torch.argsort(cell, stable=True)
    start = torch.searchsorted(cell[by_cell], torch.arange(ncx * ncy + 1, device=dev)).cpu().numpy()
    # ---- tiles of queries, compact in the projection (sort-tile partition); all boxes go to the host in one transfer
    ntile = max(1, (Q + tile - 1) // tile)
    nstrip = max(1, int(round(ntile ** 0.5)))
    qz = z[q0:q0 + Q]
    by_x = torch.argsort(qz[:, 0])
    per_strip = (Q + nstrip - 1) // nstrip
    tiles = []
    for s0 in range(0, Q, per_strip):
        strip = by_x[s0:s0 + per_strip]
        strip = strip[torch.argsort(qz[strip, 1])]
        for t0 in range(0, int(strip.numel()), tile):
            tiles.append(torch.sort(strip[t0:t0 + tile]).values)           # local query numbers of the tile, ascending
    boxes = torch.stack([torch.cat([qz[qs].min(0).values, qz[qs].max(0).values]) for qs in tiles]).cpu().numpy()
    idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    dist = torch.empty((Q, k), dtype=torch.float64, device=dev)
    cols = torch.arange(k, device=dev)[None, :]
    n_eval = 0

    def candidates(box, R):
        """Points of the grid cells the box grown by R touches (a superset of the box), ascending global numbers."""
        c0 = min(ncx - 1, max(0, int((box[0] - R - x0) / h))); c1 = min(ncx - 1, max(0, int((box[2] + R - x0) / h)))
        r0 = min(ncy - 1, max(0, int((box[1] - R - y0) / h))); r1 = min(ncy - 1, max(0, int((box[3] + R - y0) / h)))
        parts = [by_cell[start[r * ncx + c0]:start[r * ncx + c1 + 1]] for r in range(r0, r1 + 1)]
        n = sum(int(p_.numel()) for p_ in parts)
        whole = (c0, r0, c1, r1) == (0, 0, ncx - 1, ncy - 1)
        return (torch.sort(torch.cat(parts)).values if n else by_cell[:0]), n, whole

    def search(t, R):
        """One brute-force search of tile t against its candidates; returns the device scalar of its largest k-th distance."""
        nonlocal n_eval
        qs = tiles[t]
        cand, n, whole = candidates(boxes[t], R)
        while n <= k + 1 and not whole:
            R *= 2.0
            cand, n, whole = candidates(boxes[t], R)
        qg = q0 + qs
        if n <= PRUNED_SEGMENT:
            li, ld = knn_query(x64[cand], x64[qg], k + 1)
            gi = cand[li.long()]
        else:
            # more candidates than one row-free launch takes: equal pieces in index order, k + 1 nearest of each, stable merge by
            # distance (every list is sorted by (distance, index) and the pieces ascend in index: ties still come out by index)
            npiece = (n + PRUNED_SEGMENT - 1) // PRUNED_SEGMENT
            gis, lds = [], []
            for j in range(npiece):
                piece = cand[j * n // npiece:(j + 1) * n // npiece]
                li, ld = knn_query(x64[piece], x64[qg], k + 1)
                gis.append(piece[li.long()]); lds.append(ld)
            ld, order = torch.sort(torch.cat(lds, 1), dim=1, stable=True)
            gi, ld = torch.gather(torch.cat(gis, 1), 1, order)[:, :k + 1], ld[:, :k + 1]
        n_eval += n * int(qs.numel())
        own = gi == qg[:, None]                                            # the query itself: dropped; absent (k + 1 exact duplicates
        at = torch.where(own.any(1), own.to(torch.int8).argmax(1), torch.full_like(qg, k))    # in front of it): the last column goes
        take = cols + (cols >= at[:, None]).long()
        idx[qs] = gi.gather(1, take).to(torch.int32)
        td = ld.gather(1, take)
        dist[qs] = td
        return td[:, k - 1].max(), R, whole

    # first pass at the pilot radius, no host synchronisation per tile; then ONE transfer of the tiles' largest k-th distances, and
    # the tiles whose bound does not hold are searched again with R = 1.05 x what they found (an upper bound of every true k-th
    # distance of the tile, so the second search is final by construction)
    # A tile is 512 workgroups of 8 queries - half of what the device holds at once: tiles go round-robin over a few streams.
    cur = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in range(PRUNED_STREAMS)] if len(tiles) > 1 and PRUNED_STREAMS > 1 else []

    def sweep(jobs):
        for s_ in streams:
            s_.wait_stream(cur)
        out = []
        for n_, (t, R) in enumerate(jobs):
            if streams:
                with torch.cuda.stream(streams[n_ % len(streams)]):
                    out.append(search(t, R))
            else:
                out.append(search(t, R))
        for s_ in streams:
            cur.wait_stream(s_)
        return out

    first = sweep([(t, R0) for t in range(len(tiles))])
    worst = torch.stack([f[0] for f in first]).cpu().numpy()
    redo = [t for t, f in enumerate(first) if worst[t] > f[1] and not f[2]]
    second = sweep([(t, float(worst[t]) * 1.05) for t in redo])
    if second:
        w2 = torch.stack([f[0] for f in second]).cpu().numpy()
        if any(w2[i] > second[i][1] and not second[i][2] for i in range(len(redo))):
            raise RuntimeError("knn_search_pruned: radius did not settle")      # cannot happen: the second radius holds by construction
    if stats is not None:
        stats.update(distance_evaluations=n_eval, brute_force_evaluations=Q * C, tiles=len(tiles), tiles_searched_twice=len(redo), start_radius=R0,
                     grid=[ncx, ncy])
    return idx, dist


def knn_query(points, queries, k: int, query_block: int = 8192) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest of `points` (C, P) for every row of `queries` (Q, P): (idx int32 (Q,k), dist float64 (Q,k))."""
    dev = require_gpu()
    to64 = lambda a: (torch.from_numpy(np.array(a, dtype=np.float64, order="C")) if not isinstance(a, torch.Tensor) else a.double()).to(dev).contiguous()
    x64, q64 = to64(points), to64(queries)
    C, P = x64.shape
    Q = q64.shape[0]
    assert q64.shape[1] == P and 0 < k <= C
    ldx, ldq = (C + 63) // 64 * 64, (Q + 63) // 64 * 64
    xt = torch.zeros((P, ldx), dtype=torch.float32, device=dev)
    xt[:, :C] = x64.t().float()
    qt = torch.zeros((P, ldq), dtype=torch.float32, device=dev)
    qt[:, :Q] = q64.t().float()
    idx = torch.empty((Q, k), dtype=torch.int32, device=dev)
    dist = torch.empty((Q, k), dtype=torch.float64, device=dev)
    L = _lib.lib()
    qb = _knn_query_block(L, C, Q, k, query_block)
    ws = torch.empty(int(L.vcy_knn_workspace_bytes(C, qb, k)), dtype=torch.uint8, device=dev)
    for s in range(0, Q, qb):
        n = min(qb, Q - s)
        _lib.check(L.vcy_knn_query(xt.data_ptr(), x64.data_ptr(), qt.data_ptr(), q64.data_ptr(), ldq, idx[s:s + n].data_ptr(),
                                   dist[s:s + n].data_ptr(), ws.data_ptr(), C, P, ldx, s, n, k, _stream()), "knn_query")
    return idx, dist


def balance_knn_host(dsi: np.ndarray, dist: Optional[np.ndarray], lsi: np.ndarray, groups: Optional[np.ndarray], maxl: int, k: int
                     ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Host greedy balancing loop (neighbors.py:11-140) in C++ (vcy_balance_knn_host)."""
    dsi = np.ascontiguousarray(dsi, dtype=np.int64)
    n, K = dsi.shape
    if K < k:
        raise AssertionError("sight needs to be bigger than k")
    return_distance = dist is not None
    if dist is None:
        dist = np.ones(dsi.shape, dtype=np.float64)
        dist[:, 0] = 0
    dist = np.ascontiguousarray(dist, dtype=np.float64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    g = None if groups is None else np.ascontiguousarray(groups, dtype=np.int64)
    dist_new = np.empty((n, k + 1), dtype=np.float64)
    dsi_new = np.empty((n, k + 1), dtype=np.int64)
    l = np.empty(n, dtype=np.int64)
    _lib.check(_lib.lib().vcy_balance_knn_host(dsi.ctypes.data, dist.ctypes.data, lsi.ctypes.data, None if g is None else g.ctypes.data,
                                               n, K, int(maxl), int(k), int(return_distance), dist_new.ctypes.data,
                                               dsi_new.ctypes.data, l.ctypes.data), "balance_knn")
    return dist_new, dsi_new, l


def balance_knn_device_lists(idx: torch.Tensor, dist: torch.Tensor, maxl: int, k: int, groups: Optional[np.ndarray] = None
                             ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """knn_balance (neighbors.py:143-183) on sight lists that STAY where the search left them: `idx` (C, K) int32 and `dist`
    (C, K) fp64 device tensors.  The in-degree count and the processing order are device reductions, the sequential loop reads
    an int32 host copy of `idx` only (vcy_balance_knn_host32) and the distances of the k + 1 selected entries are gathered on
    the device: host memory C * K * 4 bytes instead of the reference's C * K * 16 (its default sight is the whole dataset).
    Returns (dist_new, dsi_new, l) exactly as balance_knn_host does."""
    C, K = idx.shape
    if K < k:
        raise AssertionError("sight needs to be bigger than k")
    l0 = torch.bincount(idx.reshape(-1).long(), minlength=C)
    # np.argsort(l, kind="mergesort")[::-1]: stable ascending, reversed (ties come out by DEscending cell number)
    lsi = torch.flip(torch.sort(l0, stable=True).indices, [0]).cpu().numpy().astype(np.int64)
    host = idx.cpu().numpy()                                   # int32, the only big host array
    g = None if groups is None else np.ascontiguousarray(groups, dtype=np.int64)
    pos = np.empty((C, k + 1), dtype=np.int32)
    dsi_new = np.empty((C, k + 1), dtype=np.int64)
    l = np.empty(C, dtype=np.int64)
    _lib.check(_lib.lib().vcy_balance_knn_host32(host.ctypes.data, lsi.ctypes.data, None if g is None else g.ctypes.data, C, K, int(maxl), int(k),
                                                 pos.ctypes.data, dsi_new.ctypes.data, l.ctypes.data), "balance_knn32")
    del host
    pd = torch.from_numpy(pos).to(idx.device).long()
    got = torch.gather(dist, 1, pd.clamp(min=0))
    pad = torch.from_numpy(dsi_new == np.arange(C)[:, None]).to(idx.device) & (pd < 0)     # padded slots carry dist[el, 0] (neighbors.py:65-69)
    got = torch.where(pd >= 0, got, torch.where(pad, dist[:, :1].expand_as(got), torch.zeros_like(got)))
    got[:, 0] = 0.0                                            # column 0 is never written by the loop (stays at its initial 0)
    return got.cpu().numpy(), dsi_new, l


def choice_stream_host(n: int, size: int, p: np.ndarray, cells: int, block: int = 16384, pool_factor: float = 1.5,
                       on_block=None) -> np.ndarray:
    """``np.stack([np.random.choice(n, size=size, replace=False, p=p) for _ in range(cells)])`` - the neighbour sampling of
    estimate_transition_prob (analysis.py:1561-1564) - with the same draws from numpy's global legacy RNG and the same RNG
    state afterwards, without the per-cell trips through RandomState.choice: the uniforms of a block of cells are drawn in one
    call and vcy_choice_stream_host replays choice's rounds over them.  The uniforms of the next block are drawn by a second thread
    while a block is replayed (both release the GIL); how many a cell takes is measured on a short first block.
    on_block(out, c0, c1), if given, is called as soon as rows [c0, c1) of the result are final (the caller can start device work on
    them while the replay goes on)."""
    from concurrent.futures import ThreadPoolExecutor
    p = np.ascontiguousarray(p, dtype=np.float64)
    n, size, cells = int(n), int(size), int(cells)
    if p.shape != (n,):
        raise ValueError("'a' and 'p' must have same size")
    if abs(float(p.sum()) - 1.0) > np.sqrt(np.finfo(np.float64).eps):
        raise ValueError("probabilities do not sum to 1")
    if size > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    out = np.empty((cells, size), dtype=np.int64)
    done_total = 0
    cd, used = ctypes.c_int64(0), ctypes.c_int64(0)
    pending = np.empty(0, dtype=np.float64)                  # uniforms drawn but not consumed yet (carried to the next block)
    draws = []                                               # (RNG state before the draw, number drawn), to hand back the unused tail
    per_cell = float(size) * pool_factor                     # uniforms a cell takes: a guess, then the measured mean

    def draw(want):                                          # only this function touches the global RNG while the loop runs
        state = np.random.get_state()
        return state, np.random.random_sample(want)

    def block_cells(cells_left):                             # a short first block measures what a cell takes
        return min(int(block) if draws else min(int(block), 256), cells_left)

    def want(todo, have):
        return max(int(todo * per_cell) + size - have, size)

    def hand_back(left):                                     # leave the RNG where the per-cell calls would have left it: `left` = uniforms
        while left > 0:                                      # drawn but not consumed, counted from the end of the stream
            state, drawn = draws.pop()
            np.random.set_state(state)
            if drawn >= left:
                if drawn > left:
                    np.random.random_sample(drawn - left)
                left = 0
            else:
                left -= drawn

    with ThreadPoolExecutor(1) as ex:
        fut = ex.submit(draw, want(block_cells(cells), 0)) if cells > 0 and size > 0 else None
        try:
            while fut is not None:
                state, fresh = fut.result()
                fut = None
                draws.append((state, fresh.size))
                pool = np.concatenate([pending, fresh]) if pending.size else fresh
                pending = pool                               # (until the replay has said how many it used)
                todo = block_cells(cells - done_total) if len(draws) > 1 else min(256, int(block), cells)
                # the uniforms of the block after this one are drawn while this one is replayed
                after = cells - done_total - todo
                fut = ex.submit(draw, want(min(int(block), after), max(0, pool.size - int(todo * per_cell)))) if after > 0 else None
                _lib.check(_lib.lib().vcy_choice_stream_host(pool.ctypes.data, pool.size, p.ctypes.data, n, size, todo, out[done_total:].ctypes.data,
                                                             ctypes.byref(cd), ctypes.byref(used)), "choice_stream")
                pending = pool[used.value:]
                if on_block is not None and cd.value:
                    on_block(out, done_total, done_total + cd.value)
                done_total += cd.value
                per_cell = (used.value / cd.value * 1.02 + 0.5) if cd.value else per_cell * 2       # measured; nothing fitted: draw more
                if fut is None and done_total < cells:           # the pool fell short on what was planned as the last block
                    fut = ex.submit(draw, want(cells - done_total, pending.size))
        except BaseException:
            # a failure in the replay or in the caller's on_block (a stage-D launch): the prefetched draw must not stay consumed -
            # rewind the global RNG to just after the last uniform a finished cell took, then let the error through
            extra = 0
            if fut is not None:
                state, fresh = fut.result()
                draws.append((state, fresh.size))
                extra = fresh.size
            hand_back(pending.size + extra)
            raise
    hand_back(pending.size)
    return out


# --------------------------------------------------------------------------- stage B
_fit_ws = {}


def _fit_workspace(G: int, dev) -> torch.Tensor:
    key = (G, str(dev))
    if key not in _fit_ws:
        _fit_ws.clear()
        _fit_ws[key] = torch.empty(int(_lib.lib().vcy_fit_workspace_bytes(G)), dtype=torch.uint8, device=dev)
    return _fit_ws[key]


def fit_slope(Y: CellMatrix, X: CellMatrix) -> torch.Tensor:
    """gamma (G,) float32 = max(0, <x,y>/<x,x>) per gene (estimation.py:173-188, 267-279)."""
    assert Y.t.shape == X.t.shape and Y.dtype == X.dtype
    gamma = torch.empty(Y.G, dtype=torch.float32, device=Y.t.device)
    ws = _fit_workspace(Y.G, Y.t.device)
    _lib.check(_lib.lib().vcy_fit_slope(Y.t.data_ptr(), X.t.data_ptr(), gamma.data_ptr(), ws.data_ptr(), Y.C, Y.G, Y.ld, Y.code, _stream()), "fit_slope")
    return gamma


def fit_slope_moments(Y: CellMatrix, X: CellMatrix) -> torch.Tensor:
    """(3, G) fp64 [sum xx, sum xy, sum yy] over this matrix's cells (all-reduce these across ranks)."""
    mom = torch.empty((3, Y.G), dtype=torch.float64, device=Y.t.device)
    ws = _fit_workspace(Y.G, Y.t.device)
    _lib.check(_lib.lib().vcy_fit_slope_moments(Y.t.data_ptr(), X.t.data_ptr(), mom.data_ptr(), ws.data_ptr(), Y.C, Y.G, Y.ld, Y.code, _stream()), "fit_slope_moments")
    return mom


def fit_slope_from_moments(mom: torch.Tensor) -> torch.Tensor:
    G = mom.shape[1]
    gamma = torch.empty(G, dtype=torch.float32, device=mom.device)
    _lib.check(_lib.lib().vcy_fit_slope_from_moments(mom.contiguous().data_ptr(), gamma.data_ptr(), G, _stream()), "fit_slope_from_moments")
    return gamma


def gene_moments(Y: CellMatrix, X: CellMatrix) -> torch.Tensor:
    """(5, G) fp64 [sum x, sum y, sum xx, sum xy, sum yy] per gene over cells."""
    assert Y.t.shape == X.t.shape and Y.dtype == X.dtype
    mom = torch.empty((5, Y.G), dtype=torch.float64, device=Y.t.device)
    ws = _fit_workspace(Y.G, Y.t.device)
    _lib.check(_lib.lib().vcy_gene_moments(Y.t.data_ptr(), X.t.data_ptr(), mom.data_ptr(), ws.data_ptr(), Y.C, Y.G, Y.ld, Y.code, _stream()), "gene_moments")
    return mom


def gene_stats(M, cell_scale=None, lo=None, hi=None, cell_mask=None) -> torch.Tensor:
    """(4, G) fp64 [sum, sum of squares, count(x > 0), max] per gene over cells of x = clip(M * cell_scale[:, None], lo, hi),
    restricted to the cells where cell_mask is true.  M: CellMatrix or CountMatrix."""
    dev = M.t.device
    code = M.code
    f64 = lambda t: None if t is None else torch.as_tensor(t, device=dev).to(torch.float64).contiguous()
    cell_scale, lo, hi = f64(cell_scale), f64(lo), f64(hi)
    mask = None if cell_mask is None else torch.as_tensor(cell_mask, device=dev).to(torch.uint8).contiguous()
    out = torch.empty((4, M.G), dtype=torch.float64, device=dev)
    ws = torch.empty(int(_lib.lib().vcy_gene_stats_workspace_bytes(M.G)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().vcy_gene_stats(M.t.data_ptr(), _p(cell_scale), _p(lo), _p(hi), _p(mask), out.data_ptr(), ws.data_ptr(),
                                         M.C, M.G, M.ld, code, _stream()), "gene_stats")
    return out


def _gram_workspace(C: int, G: int, L: int, symmetric: bool, dev) -> torch.Tensor:
    return torch.empty(max(16, int(_lib.lib().vcy_gram_workspace_bytes(C, G, L, int(symmetric)))), dtype=torch.uint8, device=dev)


def col_means(X: CellMatrix) -> torch.Tensor:
    """Per-gene means over the cells in fp64 (vcy_col_means; sklearn's X.mean(axis=0) of the (cells, genes) matrix PCA centres with)."""
    mean = torch.empty(X.G, dtype=torch.float64, device=X.t.device)
    ws = _gram_workspace(X.C, X.G, 1, True, X.t.device)
    _lib.check(_lib.lib().vcy_col_means(X.t.data_ptr(), mean.data_ptr(), ws.data_ptr(), X.C, X.G, X.ld, X.code, _stream()), "col_means")
    return mean


def gram(X: CellMatrix, mean: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(G, G) fp64 Gram matrix of the (centred) genes, sum_c (X[c, i] - mean[i]) (X[c, j] - mean[j]), on the f64 matrix cores
    (vcy_gram; the covariance product of perform_PCA, analysis.py:678-702).  mean None: no centring."""
    dev = X.t.device
    m = None if mean is None else mean.to(device=dev, dtype=torch.float64).contiguous()
    assert m is None or m.numel() == X.G
    out = torch.empty((X.G, X.G), dtype=torch.float64, device=dev)
    ws = _gram_workspace(X.C, X.G, X.G, True, dev)
    _lib.check(_lib.lib().vcy_gram(X.t.data_ptr(), _p(m), out.data_ptr(), ws.data_ptr(), X.C, X.G, X.ld, X.G, X.code, _stream()), "gram")
    return out


def gram_tn(X: CellMatrix, mean: Optional[torch.Tensor], Y: torch.Tensor) -> torch.Tensor:
    """(G, L) fp64 block product (X - mean)^T Y, Y (C, L) fp64 on the device (vcy_gram_tn: the second half of the subspace
    iteration's A^T (A Z))."""
    dev = X.t.device
    assert Y.dim() == 2 and Y.shape[0] == X.C and Y.dtype == torch.float64 and Y.is_cuda
    L = int(Y.shape[1])
    if L % 2 or not Y.is_contiguous():                         # rows of Y 16-byte aligned: an even pitch
        ldy = L + (L % 2)
        Yp = torch.zeros((X.C, ldy), dtype=torch.float64, device=dev)
        Yp[:, :L] = Y
    else:
        ldy, Yp = L, Y
    m = None if mean is None else mean.to(device=dev, dtype=torch.float64).contiguous()
    out = torch.empty((X.G, L), dtype=torch.float64, device=dev)
    ws = _gram_workspace(X.C, X.G, L, False, dev)
    _lib.check(_lib.lib().vcy_gram_tn(X.t.data_ptr(), _p(m), Yp.data_ptr(), out.data_ptr(), ws.data_ptr(), X.C, X.G, L, X.ld, ldy, L, X.code,
                                      _stream()), "gram_tn")
    return out


def _gene_rows(B: torch.Tensor, G: int, ld: int, dtype=torch.float64) -> torch.Tensor:
    """(N, G) rows over the genes -> the zero-padded (N, ld) operand of gemm_nt."""
    out = torch.zeros((int(B.shape[0]), ld), dtype=dtype, device=B.device)
    out[:, :G] = B
    return out


def gemm_nt(X: CellMatrix, B, row_corr: Optional[torch.Tensor] = None, col_corr: Optional[torch.Tensor] = None, c0: float = 0.0,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i, j] = sum_g X[i, g] B[j, g] - row_corr[i] - col_corr[j] + c0 in fp64 on the f64 matrix cores (vcy_gemm_nt): the products of
    perform_PCA that contract over the genes - X is read as stored (f32 / f64 rows, no fp64 copies), centring enters through the corrections.
    B: a (N, G) fp64 tensor of gene rows (components, a transposed thin block, the mean) - padded here - or a CellMatrix over the same genes
    (the cells' Gram matrix X X^T)."""
    dev = X.t.device
    assert X.ld % (32 if X.dtype == torch.float32 else 16) == 0, "the cells-major layout pads rows to 64 elements"
    if isinstance(B, CellMatrix):               # rows over the same genes: X itself (the cells' Gram matrix) or an fp64 block (CellMatrix.from_genes_major(Z))
        assert B.G == X.G and B.ld >= X.ld and (B.dtype == X.dtype or B.dtype == torch.float64)
        Bt, N, code_b = B.t, B.C, B.code
    else:
        assert B.dim() == 2 and B.shape[1] == X.G
        Bt = _gene_rows(B.to(device=dev, dtype=torch.float64), X.G, X.ld)
        N, code_b = int(B.shape[0]), _DT[torch.float64]
    if out is None:
        out = torch.empty((X.C, N), dtype=torch.float64, device=dev)
    assert out.dtype == torch.float64 and out.shape == (X.C, N) and out.stride(1) == 1
    rc = None if row_corr is None else row_corr.to(device=dev, dtype=torch.float64).contiguous()
    cc = None if col_corr is None else col_corr.to(device=dev, dtype=torch.float64).contiguous()
    assert (rc is None or rc.numel() == X.C) and (cc is None or cc.numel() == N)
    _lib.check(_lib.lib().vcy_gemm_nt(X.t.data_ptr(), Bt.data_ptr(), _p(rc), _p(cc), float(c0), out.data_ptr(), X.C, N, X.G, X.ld, int(Bt.shape[1]),
                                      int(out.stride(0)), X.code, code_b, _stream()), "gemm_nt")
    return out


def svr_fit(x, t, C: float = 1.0, epsilon: float = 0.1, gamma: float = 1.0, tol: float = 1e-3, max_iter: int = -1):
    """epsilon-SVR (RBF kernel, scalar inputs) fitted on the device: (coef (n) = alpha - alpha*, intercept (1), info (4) int32 =
    [SMO steps, converged, barrier failed, workgroups]).  The fit sklearn.svm.SVR(C, epsilon, gamma, tol).fit(x[:, None], t) does
    with libsvm (analysis.py:280-282, 324-326, 844-851), to the solver tolerance."""
    dev = require_gpu()
    x = torch.as_tensor(x, device=dev).to(torch.float64).contiguous().ravel()
    t = torch.as_tensor(t, device=dev).to(torch.float64).contiguous().ravel()
    n = int(x.numel())
    if n < 1 or int(t.numel()) != n:
        raise ValueError("svr_fit: x and t must be non-empty vectors of the same length")
    if not (bool(torch.isfinite(x).all()) and bool(torch.isfinite(t).all())):
        raise ValueError("Input contains NaN, infinity or a value too large for dtype('float64').")        # sklearn's check_array
    coef = torch.empty(n, dtype=torch.float64, device=dev)
    intercept = torch.empty(1, dtype=torch.float64, device=dev)
    info = torch.empty(4, dtype=torch.int32, device=dev)
    ws = torch.empty(int(_lib.lib().vcy_svr_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().vcy_svr_rbf_fit(x.data_ptr(), t.data_ptr(), coef.data_ptr(), intercept.data_ptr(), info.data_ptr(), ws.data_ptr(),
                                          n, float(C), float(epsilon), float(gamma), float(tol), int(max_iter), _stream()), "svr_rbf_fit")
    return coef, intercept, info


def svr_predict(x, coef, intercept, xq, gamma: float) -> torch.Tensor:
    """SVR.predict: sum_k coef[k] exp(-gamma (xq - x[k])^2) + intercept for every query."""
    dev = require_gpu()
    f64 = lambda v: torch.as_tensor(v, device=dev).to(torch.float64).contiguous().ravel()
    x, coef, intercept, xq = f64(x), f64(coef), f64(intercept), f64(xq)
    assert x.numel() == coef.numel() and intercept.numel() == 1
    out = torch.empty(int(xq.numel()), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().vcy_svr_rbf_predict(x.data_ptr(), coef.data_ptr(), intercept.data_ptr(), xq.data_ptr(), out.data_ptr(),
                                              int(x.numel()), int(xq.numel()), float(gamma), _stream()), "svr_rbf_predict")
    return out


def select_cells(M, keep) -> "CellMatrix":
    """Cell (row) subset of a cells-major matrix (CellMatrix or CountMatrix)."""
    idx = torch.nonzero(torch.as_tensor(keep, device=M.t.device), as_tuple=False).ravel()
    return type(M)(M.t.index_select(0, idx).contiguous(), M.G)


def select_genes(M: CellMatrix, keep: torch.Tensor) -> CellMatrix:
    """Gene (column) subset of a cells-major matrix, re-padded (index plumbing)."""
    idx = torch.nonzero(torch.as_tensor(keep, device=M.t.device), as_tuple=False).ravel()
    out = type(M)(torch.zeros((M.C, padded_ld(int(idx.numel()))), dtype=M.t.dtype, device=M.t.device), int(idx.numel()))
    out.t[:, : idx.numel()] = M.t.index_select(1, idx)
    return out


def gene_quantiles(M: CellMatrix, qs: Sequence[float], M2: Optional[CellMatrix] = None, scale_a: Optional[torch.Tensor] = None,
                   scale_b: Optional[torch.Tensor] = None, mask_src: Optional[CellMatrix] = None,
                   mask_thr: Optional[torch.Tensor] = None, mask_mode: int = 0) -> torch.Tensor:
    """np.percentile(M_or_Z, qs, axis=cells) -> (len(qs), G) float64 on device.
    Z = M/scale_a + M2/scale_b when scales are given; mask_mode 1/2 restricts each gene to the cells
    with mask_src > / <= mask_thr[g] (conditional percentiles)."""
    qs = np.ascontiguousarray(qs, dtype=np.float64).ravel()
    dev = M.t.device
    out = torch.empty((len(qs), M.G), dtype=torch.float64, device=dev)
    ws = torch.empty(int(_lib.lib().vcy_quantile_workspace_bytes(M.C, M.G)) // (2 if M.code == F32 else 1), dtype=torch.uint8, device=dev)
    f64 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float64).contiguous()
    scale_a, scale_b, mask_thr = f64(scale_a), f64(scale_b), f64(mask_thr)
    _lib.check(_lib.lib().vcy_gene_quantiles(M.t.data_ptr(), None if M2 is None else M2.t.data_ptr(), _p(scale_a), _p(scale_b),
                                             None if mask_src is None else mask_src.t.data_ptr(), _p(mask_thr), int(mask_mode),
                                             qs.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(qs), out.data_ptr(), ws.data_ptr(),
                                             M.C, M.G, M.ld, M.code, _stream()), "gene_quantiles")
    return out


def fit_weighted(Y: CellMatrix, X: CellMatrix, weight_mode: int, W: Optional[CellMatrix] = None, M: Optional[CellMatrix] = None,
                 M2: Optional[CellMatrix] = None, scale_a=None, scale_b=None, down=None, up=None, fit_offset: bool = True,
                 box_q: bool = True, lo_gamma: float = 1e-8, up_gamma_default: float = 20.0, up_gamma=None, q_fixed=None,
                 want_R2: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    dev = Y.t.device
    G = Y.G
    gamma = torch.empty(G, dtype=torch.float32, device=dev)
    q = torch.empty(G, dtype=torch.float32, device=dev)
    R2 = torch.empty(G, dtype=torch.float32, device=dev) if want_R2 else None
    ws = _fit_workspace(G, dev)
    f64 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float64).contiguous()
    scale_a, scale_b, down, up, up_gamma, q_fixed = map(f64, (scale_a, scale_b, down, up, up_gamma, q_fixed))
    _lib.check(_lib.lib().vcy_fit_weighted(Y.t.data_ptr(), X.t.data_ptr(), weight_mode, None if W is None else W.t.data_ptr(),
                                           None if M is None else M.t.data_ptr(), None if M2 is None else M2.t.data_ptr(),
                                           _p(scale_a), _p(scale_b), _p(down), _p(up), int(fit_offset), int(box_q), float(lo_gamma),
                                           float(up_gamma_default), _p(up_gamma), _p(q_fixed), gamma.data_ptr(), q.data_ptr(), _p(R2),
                                           ws.data_ptr(), Y.C, G, Y.ld, Y.code, _stream()), "fit_weighted")
    return gamma, q, R2


# --------------------------------------------------------------------------- stage C
def velocity_chain(Sx_sz: CellMatrix, Ux_sz: CellMatrix, gamma: torch.Tensor, q: Optional[torch.Tensor], *, want=("dmat",),
                   eps_thr: Optional[torch.Tensor] = None, dt_shift: float = 1.0, dt_extrap: float = 1.0, used_dt: float = 1.0,
                   assumption: int = 0, clip: bool = True, transform: int = SQRT, psc: float = 1e-10) -> dict:
    """Fused predict_U -> velocity -> shift -> extrapolate -> dmat; `want` names the outputs to
    materialise among Upred, velocity, delta_S, Sx_sz_t, dmat.  Returns {name: CellMatrix}."""
    names = ("Upred", "velocity", "delta_S", "Sx_sz_t", "dmat")
    outs = {n: (CellMatrix(torch.empty_like(Sx_sz.t), Sx_sz.G) if n in want else None) for n in names}
    dev = Sx_sz.t.device
    gamma = gamma.to(device=dev, dtype=torch.float32).contiguous()
    q = None if q is None else q.to(device=dev, dtype=torch.float32).contiguous()
    eps_thr = None if eps_thr is None else eps_thr.to(device=dev, dtype=torch.float64).contiguous()
    ptr = lambda n: None if outs[n] is None else outs[n].t.data_ptr()
    _lib.check(_lib.lib().vcy_velocity_chain(Sx_sz.t.data_ptr(), Ux_sz.t.data_ptr(), gamma.data_ptr(), _p(q), _p(eps_thr), ptr("Upred"),
                                             ptr("velocity"), ptr("delta_S"), ptr("Sx_sz_t"), ptr("dmat"), Sx_sz.C, Sx_sz.G, Sx_sz.ld,
                                             float(dt_shift), float(dt_extrap), float(used_dt), int(assumption), int(clip), int(transform),
                                             float(psc), Sx_sz.code, _stream()), "velocity_chain")
    return {n: m for n, m in outs.items() if m is not None}


def lincomb(x: CellMatrix, y: Optional[CellMatrix], a: float, b: float = 0.0, zero_below: Optional[torch.Tensor] = None,
            clip: bool = False) -> CellMatrix:
    """a * x + b * y (y may be None), |.| < zero_below[g] -> 0, optionally clipped at 0 (vcy_lincomb): one stage of the
    velocity chain from its stored predecessor."""
    assert y is None or (y.t.shape == x.t.shape and y.dtype == x.dtype)
    out = CellMatrix(torch.empty_like(x.t), x.G)
    zb = None if zero_below is None else zero_below.to(device=x.t.device, dtype=torch.float64).contiguous()
    _lib.check(_lib.lib().vcy_lincomb(x.t.data_ptr(), None if y is None else y.t.data_ptr(), out.t.data_ptr(), float(a), float(b), _p(zb),
                                      int(clip), x.C, x.G, x.ld, x.code, _stream()), "lincomb")
    return out


# --------------------------------------------------------------------------- pre-step + E/F helpers
def row_sums(M: CellMatrix) -> torch.Tensor:
    """cell sizes: M.sum over genes per cell (S.sum(0) in the reference layout) -> (C,) fp64."""
    out = torch.empty(M.C, dtype=torch.float64, device=M.t.device)
    _lib.check(_lib.lib().vcy_row_sums(M.t.data_ptr(), out.data_ptr(), M.C, M.G, M.ld, M.code, _stream()), "row_sums")
    return out


def scale_log(M: CellMatrix, factor: Optional[torch.Tensor], want_sz: bool = True, want_norm: bool = True, pcount: float = 1.0,
              fix_nonfinite: bool = False) -> Tuple[Optional[CellMatrix], Optional[CellMatrix]]:
    """(factor[c] * M, log2(factor[c] * M + pcount)) -- analysis.py:549-551, 579-582."""
    dev = M.t.device
    sz = CellMatrix(torch.empty_like(M.t), M.G) if want_sz else None
    nm = CellMatrix(torch.empty_like(M.t), M.G) if want_norm else None
    factor = None if factor is None else factor.to(device=dev, dtype=torch.float64).contiguous()
    _lib.check(_lib.lib().vcy_scale_log(M.t.data_ptr(), _p(factor), None if sz is None else sz.t.data_ptr(),
                                        None if nm is None else nm.t.data_ptr(), M.C, M.G, M.ld, float(pcount), int(fix_nonfinite),
                                        M.code, _stream()), "scale_log")
    return sz, nm


def delta_transform(hi_dim: CellMatrix, delta_S: CellMatrix, used_dt: float, mode: int, psc: float,
                    out: Optional[CellMatrix] = None) -> Tuple[CellMatrix, Optional[CellMatrix]]:
    """dmat (and e = log2(hi_dim + psc) for mode 3 = logratio) from a stored delta_S."""
    assert hi_dim.t.shape == delta_S.t.shape and hi_dim.dtype == delta_S.dtype
    assert out is None or (out.t.shape == hi_dim.t.shape and out.dtype == hi_dim.dtype and out.t.is_contiguous())
    dm = CellMatrix(torch.empty_like(hi_dim.t), hi_dim.G) if out is None else out
    eo = CellMatrix(torch.empty_like(hi_dim.t), hi_dim.G) if mode == 3 else None
    _lib.check(_lib.lib().vcy_delta_transform(hi_dim.t.data_ptr(), delta_S.t.data_ptr(), dm.t.data_ptr(), None if eo is None else eo.t.data_ptr(),
                                              hi_dim.C, hi_dim.G, hi_dim.ld, float(used_dt), int(mode), float(psc), hi_dim.code, _stream()),
               "delta_transform")
    return dm, eo


def permute_rows_nsign(delta_S: CellMatrix, seed: int, gene_major: Optional[bool] = None,
                       scratch: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> CellMatrix:
    """The randomised control's delta_S (analysis.py:2407-2420): per gene, the values shuffled across the cells by an independent
    pseudo-random permutation and multiplied by independent random signs (vcy_permute_rows_nsign; a function of (seed, gene, cell),
    statistical parity with the reference's numba stream).  gene_major: shuffle on a gene-major copy (two matrix-sized scratch
    buffers, three streaming passes) instead of one gather with a sector per value; None = when the matrix is large and the scratch
    is given or fits in the free memory.  scratch: two device buffers of at least the matrix's size to use for it (their contents
    are destroyed) - e.g. the buffers the caller is about to fill anyway.  Same result either way."""
    L = _lib.lib()
    out = CellMatrix(torch.empty_like(delta_S.t), delta_S.G)
    need = int(L.vcy_permute_rows_nsign_workspace_bytes(delta_S.C, delta_S.G, delta_S.code))
    if scratch is not None:
        assert all(t.is_contiguous() and t.numel() * t.element_size() >= need and t.device == delta_S.t.device for t in scratch)
    if gene_major is None:
        gene_major = delta_S.C * delta_S.G >= (1 << 24) and delta_S.G <= 65535 and \
            (scratch is not None or torch.cuda.mem_get_info(delta_S.t.device)[0] > 4 * need)
    a = b = None
    if gene_major:
        a, b = scratch if scratch is not None else (torch.empty(need, dtype=torch.uint8, device=delta_S.t.device) for _ in range(2))
    _lib.check(L.vcy_permute_rows_nsign(delta_S.t.data_ptr(), out.t.data_ptr(), _p(a), _p(b), delta_S.C, delta_S.G, delta_S.ld,
                                        int(seed) & (2**64 - 1), delta_S.code, _stream()), "permute_rows_nsign")
    return out


def corr_fixup(vals: torch.Tensor, ixs: torch.Tensor, cell0: int = 0, zero_self: bool = True, fix_nan: bool = True,
               nan_to: float = 1.0) -> int:
    """In place: zero the self pairs, NaN -> nan_to; returns the number of NaNs met (host sync)."""
    ix = _as_i32(ixs, vals.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=vals.device)
    _lib.check(_lib.lib().vcy_corr_fixup(vals.data_ptr(), ix.data_ptr(), cell0, vals.shape[0], vals.shape[1], int(zero_self), int(fix_nan),
                                         float(nan_to), cnt.data_ptr(), _DT[vals.dtype], _stream()), "corr_fixup")
    return int(cnt.item())


def transition_prob(corr: torch.Tensor, ixs: torch.Tensor, embedding, sigma_corr: float, cell0: int = 0,
                    want_tp: bool = True, want_wdiff: bool = True):
    """(tp, wdiff, delta_embedding) in neighbour-list form -- see vcy_transition_prob."""
    dev = corr.device
    ix = _as_i32(ixs, dev)
    emb = (torch.from_numpy(np.ascontiguousarray(embedding, dtype=np.float64)) if not isinstance(embedding, torch.Tensor) else embedding.double()).to(dev).contiguous()
    C_out, n = corr.shape
    tp = torch.empty_like(corr) if want_tp else None
    wd = torch.empty_like(corr) if want_wdiff else None
    de = torch.empty((C_out, emb.shape[1]), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().vcy_transition_prob(corr.contiguous().data_ptr(), ix.data_ptr(), emb.data_ptr(), emb.shape[1], _p(tp), _p(wd),
                                              de.data_ptr(), cell0, C_out, n, float(sigma_corr), _DT[corr.dtype], _stream()), "transition_prob")
    return tp, wd, de


def row_cosproj(A: CellMatrix, B: CellMatrix) -> torch.Tensor:
    out = torch.empty(A.C, dtype=torch.float64, device=A.t.device)
    _lib.check(_lib.lib().vcy_row_cosproj(A.t.data_ptr(), B.t.data_ptr(), out.data_ptr(), A.C, A.G, A.ld, A.code, _stream()), "row_cosproj")
    return out


def embedding_scaling(hi: CellMatrix, dS: CellMatrix, ixs, wdiff: torch.Tensor, dS_rndm: Optional[CellMatrix] = None,
                      wdiff_rndm: Optional[torch.Tensor] = None, order: Optional[torch.Tensor] = None, validate: bool = True):
    """cos_proj (C_out) fp64 [and the control's] of calculate_embedding_shift's expression scaling (analysis.py:1714-1719, 1726-1731) in one
    launch, the (genes, cells) estimates never written (vcy_embedding_scaling).  Returns None when the neighbour lists are wider than
    the kernel sorts in one workgroup - the caller then pools with knn_pool[_w2] + row_cosproj."""
    dev = hi.t.device
    ix = _as_i32(ixs, dev)
    C_out, n = ix.shape
    L = _lib.lib()
    if n > int(L.vcy_embedding_scaling_max_neighbors()) or n == 0 or C_out == 0:
        return None
    assert dS.t.shape == hi.t.shape and dS.dtype == hi.dtype and C_out <= hi.C
    if validate and (int(ix.min()) < 0 or int(ix.max()) >= hi.C):
        raise ValueError("neighbour index out of range")
    w = wdiff.to(device=dev, dtype=hi.dtype).contiguous()
    assert tuple(w.shape) == (C_out, n)
    dual = dS_rndm is not None
    w2 = None
    if dual:
        assert wdiff_rndm is not None and dS_rndm.t.shape == hi.t.shape and dS_rndm.dtype == hi.dtype
        w2 = wdiff_rndm.to(device=dev, dtype=hi.dtype).contiguous()
    cos = torch.empty(C_out, dtype=torch.float64, device=dev)
    cos2 = torch.empty(C_out, dtype=torch.float64, device=dev) if dual else None
    order, n_sched = _sched(order, dev, C_out)
    assert n_sched == C_out, "embedding_scaling: the schedule must cover every cell"
    _lib.check(L.vcy_embedding_scaling(hi.t.data_ptr(), dS.t.data_ptr(), None if not dual else dS_rndm.t.data_ptr(), ix.data_ptr(), w.data_ptr(),
                                       _p(w2), _p(order), cos.data_ptr(), _p(cos2), hi.C, hi.G, hi.ld, C_out, n, hi.code, _stream()), "embedding_scaling")
    return (cos, cos2) if dual else (cos,)


def _run_steps(step, x: torch.Tensor, y: torch.Tensor, n_steps: int) -> torch.Tensor:
    """n_steps of step(src, dst) ping-ponging between x and y; returns the buffer that holds the last iterate.  Long loops are
    launch-bound (a few tiny kernels per step, thousands of steps): an x -> y -> x pair of steps is captured into a hipGraph once
    and replayed (the kernels take raw pointers and never allocate or sync)."""
    done = 0
    if n_steps >= 32:
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(x, y); step(y, x)                      # warm-up outside capture
            # captured by hand: the torch.cuda.graph context empties the caching allocator on entry, and every later method of
            # the pipeline would then pay for fresh device allocations (24 GB for the next normalize: 0.5 s); the steps allocate
            # nothing, so there is no pool to keep tidy
            graph = torch.cuda.CUDAGraph()
            graph.capture_begin()
            try:
                step(x, y); step(y, x)
            finally:
                graph.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        done = 2
        for _ in range((n_steps - done) // 2):
            graph.replay()
        done += 2 * ((n_steps - done) // 2)
    for _ in range(n_steps - done):
        step(x, y)
        x, y = y, x
    return x


def diffuse(x0, tr, n_steps: int, accumulate: bool) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """n_steps of x <- x . tr on device.  tr: dense torch (n,n) f32/f64, a scipy sparse matrix, or the MarkovFactors of
    prepare_markov_factored.  Returns (x_final, sum of the iterates if accumulate)."""
    import scipy.sparse as sp
    dev = require_gpu()
    L = _lib.lib()
    x = (torch.from_numpy(np.ascontiguousarray(x0, dtype=np.float64)) if not isinstance(x0, torch.Tensor) else x0.double()).to(dev).contiguous().clone()
    n = x.numel()
    y = torch.empty_like(x)
    acc = torch.zeros_like(x) if accumulate else None
    if isinstance(tr, MarkovFactors):
        assert tr.n == n
        ws = torch.empty(int(L.vcy_markov_factored_workspace_bytes(n)), dtype=torch.uint8, device=dev)

        chain = {"prepared": 0}                           # after the first step every step starts from the previous one's output

        def step(src, dst):
            prepared, chain["prepared"] = chain["prepared"], 1
            if tr.cull is not None:
                es_sorted, rank, boxes, cut, order = tr.cull
                _lib.check(L.vcy_diffuse_step_factored_culled(src.data_ptr(), dst.data_ptr(), _p(acc), tr.colptr.data_ptr(), tr.rowidx.data_ptr(),
                                                              tr.scsc.data_ptr(), tr.tot.data_ptr(), tr.kw.data_ptr(), es_sorted.data_ptr(), rank.data_ptr(),
                                                              order.data_ptr(), boxes.data_ptr(), tr.edim, tr.sigma_W, cut, ws.data_ptr(), n, prepared,
                                                              _DT[tr.compute_dtype], _stream()), "diffuse_step_factored_culled")
                return
            _lib.check(L.vcy_diffuse_step_factored(src.data_ptr(), dst.data_ptr(), _p(acc), tr.colptr.data_ptr(), tr.rowidx.data_ptr(), tr.scsc.data_ptr(),
                                                   tr.tot.data_ptr(), tr.kw.data_ptr(), tr.es.data_ptr(), tr.edim, tr.sigma_W, ws.data_ptr(), n, prepared,
                                                   _DT[tr.compute_dtype], _stream()), "diffuse_step_factored")
        x = _run_steps(step, x, y, n_steps)
    elif sp.issparse(tr):
        csc = sp.csc_matrix(tr)
        csc.sort_indices()
        colptr = torch.from_numpy(csc.indptr.astype(np.int64)).to(dev)
        rowidx = torch.from_numpy(csc.indices.astype(np.int32)).to(dev)
        val = torch.from_numpy(np.ascontiguousarray(csc.data, dtype=np.float64)).to(dev)

        def step(src, dst):
            _lib.check(L.vcy_diffuse_step_csc(colptr.data_ptr(), rowidx.data_ptr(), val.data_ptr(), src.data_ptr(), dst.data_ptr(), _p(acc), n, F64, _stream()), "diffuse_step_csc")
        x = _run_steps(step, x, y, n_steps)
    else:
        T = tr.to(dev).contiguous()
        assert T.shape == (n, n) and T.dtype in _DT
        ws = torch.empty(int(L.vcy_diffuse_workspace_bytes(n)), dtype=torch.uint8, device=dev)

        def step(src, dst):
            _lib.check(L.vcy_diffuse_step_dense(T.data_ptr(), src.data_ptr(), dst.data_ptr(), _p(acc), ws.data_ptr(), n, _DT[T.dtype], _stream()), "diffuse_step_dense")

        x = _run_steps(step, x, y, n_steps)
    return x, acc


def gamma_weights(S: CellMatrix, U: Optional[CellMatrix], mode: int, pa, pb, pc=None, pd=None, sa=None, sb=None, power: float = 15.0) -> CellMatrix:
    """Dense W for the non-default fit_gammas weight modes (see vcy_gamma_weights)."""
    dev = S.t.device
    f64 = lambda t: None if t is None else t.to(device=dev, dtype=torch.float64).contiguous()
    pa, pb, pc, pd, sa, sb = map(f64, (pa, pb, pc, pd, sa, sb))
    W = CellMatrix(torch.empty_like(S.t), S.G)
    _lib.check(_lib.lib().vcy_gamma_weights(S.t.data_ptr(), None if U is None else U.t.data_ptr(), W.t.data_ptr(), _p(pa), _p(pb), _p(pc), _p(pd),
                                            _p(sa), _p(sb), S.C, S.G, S.ld, int(mode), float(power), S.code, _stream()), "gamma_weights")
    return W


class MarkovFactors:
    """The Markov chain of prepare_markov (analysis.py:1853-1862) without its dense (n, n) matrix:
    tr[c, j] = (0.2 K_W(c, j) / kw[c] + s[c, j]) / tot[c], s sparse (CSC, diagonal included), K_W the Gaussian of the embedding
    distance - what vcy_diffuse_step_factored steps through.  `dense()` materialises tr (vcy_prepare_markov) when it is asked for."""

    def __init__(self, csr, embedding, sigma_D, sigma_W, colptr, rowidx, scsc, tot, kw, es, compute_dtype):
        self._csr, self.embedding, self.sigma_D, self.sigma_W = csr, embedding, float(sigma_D), float(sigma_W)
        self.colptr, self.rowidx, self.scsc, self.tot, self.kw, self.es, self.compute_dtype = colptr, rowidx, scsc, tot, kw, es, compute_dtype
        self.n, self.edim = int(embedding.shape[0]), int(embedding.shape[1])
        self.shape = (self.n, self.n)
        self.cull = None                                    # (es_sorted, rank, boxes, cut, order) of the culled Gauss transform, see enable_culling

    def enable_culling(self, cut: Optional[float] = None) -> "MarkovFactors":
        """Sort the cells along the Hilbert curve of the embedding and box runs of them, so that the steps skip source runs whose
        contribution to a block of targets is below 2^-cut of their weight (vcy_diffuse_step_factored_culled).  Pays when sigma_W
        is small against the extent of the embedding, costs a few per cent when it is not."""
        dev = self.es.device
        if cut is None:
            cut = 48.0 if self.compute_dtype == torch.float32 else 72.0
        if self.edim >= 2:
            order = hilbert_order(self.embedding[:, :2].contiguous()).long()
        else:
            order = torch.argsort(self.embedding[:, 0])
        rank = torch.empty(self.n, dtype=torch.int32, device=dev)
        rank[order] = torch.arange(self.n, dtype=torch.int32, device=dev)
        es_sorted = self.es.index_select(0, order).contiguous()
        code = _DT[self.compute_dtype]
        boxes = torch.empty(int(_lib.lib().vcy_markov_cull_boxes_bytes(self.n, self.edim, code)), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().vcy_markov_cull_boxes(es_sorted.data_ptr(), boxes.data_ptr(), self.n, self.edim, code, _stream()), "markov_cull_boxes")
        self.cull = (es_sorted, rank, boxes, float(cut), order.to(torch.int32).contiguous())
        return self

    def dense(self, dtype=torch.float64) -> torch.Tensor:
        ip, ix, pv = self._csr
        return prepare_markov(ip, ix, pv, self.embedding, self.sigma_D, self.sigma_W, dtype=dtype)


def prepare_markov_factored(indptr, indices, pval, embedding, sigma_D: float, sigma_W: float, compute_dtype=torch.float32,
                            cull: Optional[bool] = None) -> MarkovFactors:
    """Factors of the Markov matrix from CSR transition probabilities (vcy_prepare_markov_factored); O(nnz + n) memory.
    cull: step with the culled Gauss transform (MarkovFactors.enable_culling); None = when the embedding is wider than twice the
    radius beyond which the kernel is dropped (else every box is in range of every other and the tests only cost)."""
    dev = require_gpu()
    ip = torch.as_tensor(np.ascontiguousarray(indptr, dtype=np.int64)).to(dev) if not isinstance(indptr, torch.Tensor) else indptr.to(dev, torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    pv = (torch.as_tensor(np.ascontiguousarray(pval, dtype=np.float64)) if not isinstance(pval, torch.Tensor) else pval.double()).to(dev).contiguous()
    emb = (torch.from_numpy(np.ascontiguousarray(embedding, dtype=np.float64)) if not isinstance(embedding, torch.Tensor) else embedding.double()).to(dev).contiguous()
    n, edim = int(emb.shape[0]), int(emb.shape[1])
    nnz = int(ix.numel())
    sval = torch.empty(nnz, dtype=torch.float64, device=dev)
    sdiag, kw, tot = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    es = torch.empty((n, edim), dtype=compute_dtype, device=dev)
    _lib.check(_lib.lib().vcy_prepare_markov_factored(ip.data_ptr(), ix.data_ptr(), pv.data_ptr(), emb.data_ptr(), edim, sval.data_ptr(), sdiag.data_ptr(),
                                                      kw.data_ptr(), tot.data_ptr(), es.data_ptr(), n, float(sigma_D), float(sigma_W),
                                                      _DT[compute_dtype], _stream()), "prepare_markov_factored")
    # s in CSC form, diagonal included (index plumbing: one sort of the nnz + n coordinates by (column, row))
    rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), ip[1:] - ip[:-1])
    cols = ix.to(torch.int64)
    keep = rows != cols                                     # a diagonal entry stored by P is replaced by the row maximum
    ar = torch.arange(n, device=dev, dtype=torch.int64)
    rows, cols, vals = torch.cat([rows[keep], ar]), torch.cat([cols[keep], ar]), torch.cat([sval[keep], sdiag])
    order = torch.argsort(cols * n + rows)
    colptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    colptr[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
    fac = MarkovFactors((ip, ix, pv), emb, sigma_D, sigma_W, colptr, rows[order].to(torch.int32).contiguous(), vals[order].contiguous(),
                        tot, kw, es, compute_dtype)
    if cull is None:
        cut = 48.0 if compute_dtype == torch.float32 else 72.0
        extent = float((es.max(0).values - es.min(0).values).max()) if n > 1 else 0.0
        cull = n >= 4096 and extent > 2.0 * cut ** 0.5
    if cull:
        fac.enable_culling()
    return fac


def prepare_markov(indptr, indices, pval, embedding, sigma_D: float, sigma_W: float, dtype=torch.float64) -> torch.Tensor:
    """Dense (n, n) Markov matrix on device from CSR transition probabilities (vcy_prepare_markov)."""
    dev = require_gpu()
    ip = torch.as_tensor(np.ascontiguousarray(indptr, dtype=np.int64)).to(dev) if not isinstance(indptr, torch.Tensor) else indptr.to(dev, torch.int64).contiguous()
    ix = _as_i32(indices, dev)
    pv = (torch.as_tensor(np.ascontiguousarray(pval, dtype=np.float64)) if not isinstance(pval, torch.Tensor) else pval.double()).to(dev).contiguous()
    emb = (torch.from_numpy(np.ascontiguousarray(embedding, dtype=np.float64)) if not isinstance(embedding, torch.Tensor) else embedding.double()).to(dev).contiguous()
    n = emb.shape[0]
    tr = torch.empty((n, n), dtype=dtype, device=dev)
    _lib.check(_lib.lib().vcy_prepare_markov(ip.data_ptr(), ix.data_ptr(), pv.data_ptr(), emb.data_ptr(), emb.shape[1], tr.data_ptr(), n,
                                             float(sigma_D), float(sigma_W), _DT[dtype], _stream()), "prepare_markov")
    return tr


def morton_order(points, dims: int = 2) -> torch.Tensor:
    """Z-order (Morton) permutation of the rows of `points` over their first `dims` (2 or 3) coordinates:
    a cheap locality sort used only to SCHEDULE cells (kernels give identical results in any order)."""
    dev = require_gpu()
    p = (torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)) if not isinstance(points, torch.Tensor) else points).to(dev)
    p = p[:, :dims].double()
    lo, hi = p.min(0).values, p.max(0).values
    bits = 16 if dims == 2 else 10
    q = ((p - lo) / (hi - lo + 1e-300) * (2 ** bits - 1)).long()
    code = torch.zeros(p.shape[0], dtype=torch.int64, device=dev)
    for b in range(bits):
        for d in range(dims):
            code |= ((q[:, d] >> b) & 1) << (b * dims + d)
    return torch.argsort(code).to(torch.int32)


def hilbert_order(points, bits: int = 16) -> torch.Tensor:
    """Hilbert-curve permutation of the rows of `points` over their first two coordinates.  Unlike the Z-order curve it
    has no jumps: consecutive cells are always spatial neighbours, so 8-cell groups of the stage-D schedule share more
    of their sampled neighbours.  Scheduling only (kernels give identical results in any order)."""
    dev = require_gpu()
    p = (torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)) if not isinstance(points, torch.Tensor) else points).to(dev)
    p = p[:, :2].double()
    lo, hi = p.min(0).values, p.max(0).values
    q = ((p - lo) / (hi - lo + 1e-300) * (2 ** bits - 1)).long()
    x, y = q[:, 0].clone(), q[:, 1].clone()
    d = torch.zeros_like(x)
    s = 1 << (bits - 1)
    while s > 0:                                      # the classic xy -> d walk, vectorised over the cells
        rx = ((x & s) > 0).long()
        ry = ((y & s) > 0).long()
        d += s * s * ((3 * rx) ^ ry)
        flip = (ry == 0) & (rx == 1)                  # rotate the quadrant
        x = torch.where(flip, s - 1 - x, x)
        y = torch.where(flip, s - 1 - y, y)
        swap = ry == 0
        x, y = torch.where(swap, y, x), torch.where(swap, x, y)
        s >>= 1
    return torch.argsort(d).to(torch.int32)


class ClockProbe:
    """Shader clock while other kernels run (vcy_clock_probe; measurement only).  `start(duration_ms)` launches the one-wave-per-XCD
    probe on a side stream; after the kernels of interest have been enqueued and the device synchronised, `ghz()` returns
    (mean, min, max) of the clock over the sampling intervals, in GHz."""

    def __init__(self, nblocks: int = 8, interval_ms: float = 2.0):
        self.dev = require_gpu()
        self.nblocks, self.interval = nblocks, int(interval_ms * 1e5)       # ticks of the 100 MHz counter
        self.stream = torch.cuda.Stream(device=self.dev)
        self.samples = None

    def start(self, duration_ms: float) -> None:
        # (the kernel refuses nsamples x interval beyond 3e8 ticks = 3 s of the 100 MHz counter: a longer request samples its first 3 s)
        n = int(min(4096, 300_000_000 // max(self.interval, 1), max(2, duration_ms * 1e5 / self.interval + 1)))
        self.samples = torch.zeros((self.nblocks, n, 2), dtype=torch.int64, device=self.dev)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))     # the zero fill runs on the current stream: the probe starts after it
        _lib.check(_lib.lib().vcy_clock_probe(self.samples.data_ptr(), self.nblocks, n, self.interval, self.stream.cuda_stream), "clock_probe")

    def ghz(self) -> Tuple[float, float, float]:
        self.stream.synchronize()
        s = self.samples.cpu().numpy().astype(np.float64)
        dc, dr = np.diff(s[:, :, 0], axis=1), np.diff(s[:, :, 1], axis=1)
        f = dc / np.maximum(dr, 1.0) * 0.1
        return float(f.mean()), float(f.min()), float(f.max())
