"""CPU oracle for the velocyto.py analysis hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  The product
(``velocyto.py_amd/``) never imports it.

What it is: a NumPy/SciPy + C (``velocyto_oracle.c``) restatement, in fp64 and in the
reference's own ``(genes, cells)`` layout, of every function on the path SURVEY.md section 8(a)
lists.  Each function cites the reference lines it follows (paths relative to
``/root/reference/velocyto/``).

Parity pin: the reference has no tests/golden vectors of its own; this oracle is pinned
against the reference itself (its Cython kernels built by ``oracle/build_ref.py`` and its
Python modules imported from /root/reference) by ``tests/golden/make_golden.py``; the
resulting vectors live in ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` replays
them everywhere (also on the GPU box, where /root/reference does not exist).

Third-party arithmetic the reference delegates to (all unpinned in its setup.py:39-52):
  * scikit-learn ``NearestNeighbors`` (exact kNN)  -> restated as brute-force fp64 kNN,
    ties broken by index (``knn_search``);
  * ``scipy.optimize.nnls`` on one column          -> closed form max(0, <x,y>/<x,x>);
  * ``scipy.optimize.minimize(L-BFGS-B)`` / ``minimize_scalar(bounded)`` / ``leastsq``
    -> called exactly as the reference calls them (SciPy 1.15.3 is in the image on both
    boxes) AND restated as exact closed-form box-constrained least squares
    (``*_exact``), which is what the GPU path implements;
  * ``numpy.percentile`` (linear interpolation), ``numpy.random`` legacy stream -> used as is;
  * scikit-learn ``SVR`` (libsvm 3.x epsilon-SVR, RBF kernel; analysis.py:280-282, 324-326, 844-851) -> called as the reference
    calls it (scikit-learn 1.7.2 is in the image on both boxes) AND restated as libsvm's published iteration (SMO with
    second-order working-set selection, Fan, Chen & Lin 2005) in ``svr_rbf_fit``, which is what the GPU path implements;
    ``tests/test_oracle_golden.py`` pins the restatement on scikit-learn.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np
import scipy.optimize
from scipy import sparse

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvelocyto_oracle.so")
_lib = None

LINEAR, SQRT, LOG10 = 0, 1, 2
_TRANSFORMS = {"linear": LINEAR, "sqrt": SQRT, "log10": LOG10, "log": LOG10}


def build(force: bool = False) -> str:
    """Compile velocyto_oracle.c -> libvelocyto_oracle.so (gcc, OpenMP)."""
    src = os.path.join(_HERE, "velocyto_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-fopenmp", "-std=c11", "-shared", src, "-o", _LIB_PATH, "-lm"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int64)
        fp = ctypes.POINTER(ctypes.c_float)
        ci, cd = ctypes.c_int, ctypes.c_double
        L.vo_max_threads.restype = ci
        L.vo_coldeltacor.argtypes = [dp, dp, dp, ci, ci, ci, cd, ci]
        L.vo_coldeltacor_partial.argtypes = [dp, dp, dp, ip, ci, ci, ci, ci, cd, ci]
        L.vo_coldeltacor_partial_compact.argtypes = [dp, dp, dp, ip, ci, ci, ci, ci, cd, ci, ci, ci]
        L.vo_convolve_csr.argtypes = [dp, ip, ip, dp, dp, ci, ci, ci]
        L.vo_fit_slope.argtypes = [dp, dp, fp, ci, ci]
        L.vo_balance_knn.argtypes = [ip, dp, ip, ip, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int64, ci, dp, ip, ip]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def max_threads() -> int:
    return int(lib().vo_max_threads())


# --------------------------------------------------------------------------- correlations
def coldeltacor(emat, dmat, transform="linear", psc=0.0, threads=0) -> np.ndarray:
    """estimation.colDeltaCor / colDeltaCorSqrt / colDeltaCorLog10 (estimation.py:11-33, 65-87,
    119-141) over speedboosted.pyx:13-257: dense (C,C) fp64, out allocated as zeros and
    accumulated into."""
    e, d = _c64(emat), _c64(dmat)
    G, C = e.shape
    out = np.zeros((C, C))
    lib().vo_coldeltacor(_dp(e), _dp(d), _dp(out), G, C, _TRANSFORMS[transform], float(psc), int(threads))
    return out


def coldeltacor_partial(emat, dmat, ixs, transform="linear", psc=0.0, threads=0) -> np.ndarray:
    """estimation.colDeltaCor*partial (estimation.py:36-62, 90-116, 144-170) over
    speedboosted.pyx:263-538."""
    e, d = _c64(emat), _c64(dmat)
    ix = np.ascontiguousarray(ixs, dtype=np.int64)
    G, C = e.shape
    out = np.zeros((C, C))
    lib().vo_coldeltacor_partial(_dp(e), _dp(d), _dp(out), _ip(ix), G, C, ix.shape[1],
                                 _TRANSFORMS[transform], float(psc), int(threads))
    return out


def coldeltacor_partial_compact(emat, dmat, ixs, transform="linear", psc=0.0, threads=0,
                                c0=0, c1=None) -> np.ndarray:
    """Compact (C, nrndm) form of the partial kernels: out[c,n] = corr(c, ixs[c,n])."""
    e, d = _c64(emat), _c64(dmat)
    ix = np.ascontiguousarray(ixs, dtype=np.int64)
    G, C = e.shape
    c1 = C if c1 is None else c1
    out = np.zeros((C, ix.shape[1]))
    lib().vo_coldeltacor_partial_compact(_dp(e), _dp(d), _dp(out), _ip(ix), G, C, ix.shape[1],
                                         _TRANSFORMS[transform], float(psc), int(c0), int(c1), int(threads))
    return out


# --------------------------------------------------------------------------- the REAL reference kernels (oracle/_ref)
# oracle/build_ref.py compiles the reference's own Cython module (velocyto/speedboosted.pyx, its flags: -fopenmp -ffast-math) from where
# it lies under /root/reference into oracle/_ref/ - a binary that is git-ignored but travels to the GPU box with the repository
# snapshot.  Where it is present it validates the restatement above against the reference itself (tests) and is the CPU baseline of
# stage D (bench.py, cpu_baseline.kind = "reference").  It runs in a SUBPROCESS: the module is linked with -ffast-math, whose start-up
# code switches the loading process to flush-to-zero / denormals-are-zero, and its kernels take ~60 MB of scratch per thread.
_REF_KERNELS = {("sqrt", True): "_colDeltaCorSqrtpartial", ("log10", True): "_colDeltaCorLog10partial", ("linear", True): "_colDeltaCorpartial",
                ("sqrt", False): "_colDeltaCorSqrt", ("log10", False): "_colDeltaCorLog10", ("linear", False): "_colDeltaCor"}

_REF_RUNNER = r"""
import importlib.util, sys, time, numpy as np
so, work, name, threads, psc, partial = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]), sys.argv[6] == "1"
spec = importlib.util.spec_from_file_location("speedboosted", so)
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
e, d = np.load(work + "/e.npy"), np.load(work + "/d.npy")
rm = np.zeros((e.shape[1], e.shape[1]))
args = [e, d, rm]
if partial:
    ixs = np.load(work + "/ixs.npy")
    args.append(ixs)
args.append(threads)
if name not in ("_colDeltaCor", "_colDeltaCorpartial"):
    args.append(psc)
t0 = time.perf_counter()
getattr(m, name)(*args)
dt = time.perf_counter() - t0
np.save(work + "/out.npy", rm[np.arange(e.shape[1])[:, None], ixs] if partial else rm)
np.save(work + "/seconds.npy", np.array([dt]))
"""


def reference_module_status() -> Tuple[str, str]:
    """("present", path) - a module this interpreter can load; ("unloadable", why) - oracle/_ref holds a module built for another
    interpreter; ("absent", why) - nothing was built (no /root/reference where oracle/build_ref.py ran) or it did not travel."""
    import glob
    import importlib.machinery
    hits = sorted(glob.glob(os.path.join(_HERE, "_ref", "speedboosted*.so")))
    if not hits:
        return "absent", "oracle/_ref holds no speedboosted*.so (oracle/build_ref.py builds it where /root/reference exists)"
    for h in hits:
        if any(os.path.basename(h) == "speedboosted" + suf for suf in importlib.machinery.EXTENSION_SUFFIXES):
            return "present", h
    return "unloadable", f"{', '.join(os.path.basename(h) for h in hits)}: built for another interpreter than this one ({importlib.machinery.EXTENSION_SUFFIXES[0]})"


def reference_module_path() -> Optional[str]:
    """Path of the compiled reference kernel module under oracle/_ref, or None where it was not built / did not travel / cannot be loaded."""
    st, what = reference_module_status()
    return what if st == "present" else None


def reference_module_expected() -> bool:
    """Whether this tree SHOULD have loadable reference kernels: oracle/build_ref.py ran in it and left its marker
    (oracle/_ref/built.json) for THIS interpreter's extension ABI.  A tree that merely sits next to /root/reference, a stale
    oracle/_ref directory, or a module built for another Python do not make the kernels "expected": the tests that ask for them then
    skip with the reason instead of failing a correct product.  VCY_REF_OPTIONAL=1 turns the expectation off altogether."""
    import importlib.machinery
    import json
    if os.environ.get("VCY_REF_OPTIONAL") == "1":
        return False
    try:
        with open(os.path.join(_HERE, "_ref", "built.json")) as f:
            mark = json.load(f)
    except (OSError, ValueError):
        return False
    return mark.get("ext_suffix") == importlib.machinery.EXTENSION_SUFFIXES[0]


def reference_coldeltacor(emat, dmat, ixs=None, transform="linear", psc=0.0, threads=8) -> Tuple[np.ndarray, float]:
    """The reference's OWN kernel (speedboosted.pyx:13-610, as estimation.colDeltaCor* calls it: estimation.py:11-170) on
    (genes, cells) fp64 inputs, in a subprocess.  ixs given: the *partial kernel, result gathered to the compact (C, nrndm) form;
    ixs None: the full kernel, (C, C).  Returns (correlations, seconds inside the kernel call).  Raises FileNotFoundError where
    oracle/_ref holds no module."""
    import sys
    import tempfile
    so = reference_module_path()
    if so is None:
        raise FileNotFoundError("oracle/_ref holds no reference kernel module (oracle/build_ref.py builds it where /root/reference exists)")
    name = _REF_KERNELS[({"log": "log10"}.get(transform, transform), ixs is not None)]
    with tempfile.TemporaryDirectory() as work:
        np.save(os.path.join(work, "e.npy"), np.ascontiguousarray(emat, dtype=np.float64))
        np.save(os.path.join(work, "d.npy"), np.ascontiguousarray(dmat, dtype=np.float64))
        if ixs is not None:
            np.save(os.path.join(work, "ixs.npy"), np.ascontiguousarray(ixs, dtype=np.intp))
        r = subprocess.run([sys.executable, "-c", _REF_RUNNER, so, work, name, str(int(threads)), repr(float(psc)), "1" if ixs is not None else "0"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference kernel subprocess failed:\n" + r.stderr[-2000:])
        return np.load(os.path.join(work, "out.npy")), float(np.load(os.path.join(work, "seconds.npy"))[0])


_REF_RATE_RUNNER = r"""
import importlib.util, sys, time, numpy as np
so, work, name, threads, psc = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5])
spec = importlib.util.spec_from_file_location("speedboosted", so)
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
e, d = np.load(work + "/e.npy", mmap_mode="c"), np.load(work + "/d.npy", mmap_mode="c")      # copy-on-write maps: writable buffers, nothing copied
ixs = np.load(work + "/ixs.npy")
rm = np.load(work + "/rm.npy", mmap_mode="r+")
args = [e, d, rm, ixs, threads] + ([] if name == "_colDeltaCorpartial" else [psc])
np.save(work + "/started.npy", np.array([time.time()]))
t0 = time.perf_counter()
getattr(m, name)(*args)
np.save(work + "/seconds.npy", np.array([time.perf_counter() - t0]))
"""


def reference_coldeltacor_rate(work: str, ixs, transform="sqrt", psc=1e-10, threads=8, t_first=6.0, t_second=16.0) -> dict:
    """Cells per second of the reference's OWN partial kernel on a problem too large to run to the end: `work` holds e.npy and d.npy, the
    full (genes, cells) fp64 matrices (put them on a RAM-backed file system; the kernel maps them copy-on-write), `ixs` the (cells, nrndm)
    neighbour lists.  The kernel (speedboosted.pyx:352-443) has no cell range - it walks all columns under `schedule='guided'` - so it runs
    in a subprocess over ALL of them with its (cells, cells) output as a sparse file in `work`, and the rows it has finished are counted from
    outside at two instants: rate = (rows(t_second) - rows(t_first)) / (t_second - t_first) - rows in progress cancel, start-up is left out -
    and the subprocess is killed.  A problem it finishes earlier gives cells / seconds.  Returns {"cells_per_s", "rows_first", "rows_second",
    "t_first", "t_second", "threads", "finished"}."""
    import signal
    import sys
    import time
    so = reference_module_path()
    if so is None:
        raise FileNotFoundError("oracle/_ref holds no reference kernel module (oracle/build_ref.py builds it where /root/reference exists)")
    name = _REF_KERNELS[({"log": "log10"}.get(transform, transform), True)]
    ix = np.ascontiguousarray(ixs, dtype=np.intp)
    C = ix.shape[0]
    np.save(os.path.join(work, "ixs.npy"), ix)
    rm = np.lib.format.open_memmap(os.path.join(work, "rm.npy"), mode="w+", dtype=np.float64, shape=(C, C))     # sparse: pages appear as rows are written
    for f in ("started.npy", "seconds.npy"):
        if os.path.exists(os.path.join(work, f)):
            os.remove(os.path.join(work, f))
    proc = subprocess.Popen([sys.executable, "-c", _REF_RATE_RUNNER, so, work, name, str(int(threads)), repr(float(psc))], stderr=subprocess.PIPE, text=True)
    rows = np.arange(C)
    probe = ix[:, 0]
    done = lambda: int(np.count_nonzero(rm[rows, probe] != 0.0))           # (NaN != 0 too: a finished zero-variance pair counts)
    try:
        t_wait = time.time()
        while not os.path.exists(os.path.join(work, "started.npy")):
            if proc.poll() is not None:
                raise RuntimeError("reference kernel subprocess failed:\n" + proc.stderr.read()[-2000:])
            if time.time() - t_wait > 300:
                raise RuntimeError("reference kernel subprocess did not start within 300 s")
            time.sleep(0.05)
        time.sleep(0.1)
        t0 = float(np.load(os.path.join(work, "started.npy"))[0])
        out = {"threads": int(threads), "finished": False}
        # the count of finished rows, read every 0.2 s: all threads start together, so rows finish in waves - the rate is taken between the
        # FIRST and the LAST instant at which the count moved inside [t_first, t_second], not between the two ends of the window
        series = []
        while time.time() - t0 < t_second and proc.poll() is None:
            time.sleep(0.2)
            series.append((time.time() - t0, done()))
        if proc.poll() is not None and os.path.exists(os.path.join(work, "seconds.npy")):
            sec = float(np.load(os.path.join(work, "seconds.npy"))[0])
            out.update({"finished": True, "seconds": sec, "cells_per_s": C / sec, "rows_first": series[0][1] if series else C, "rows_second": C,
                        "t_first": series[0][0] if series else sec, "t_second": sec})
        elif proc.poll() is not None:
            raise RuntimeError("reference kernel subprocess failed:\n" + proc.stderr.read()[-2000:])
        else:
            moves = [(t, n) for (t, n), (_, n_prev) in zip(series[1:], series[:-1]) if n != n_prev and t >= t_first]
            if len(moves) < 2:
                raise RuntimeError(f"the reference kernel finished too few rows in {t_second:.0f} s to give a rate: {series[-1][1] if series else 0}")
            (ta, na), (tb, nb) = moves[0], moves[-1]
            out.update({"cells_per_s": (nb - na) / (tb - ta), "rows_first": na, "rows_second": nb, "t_first": ta, "t_second": tb, "readings": len(series)})
        return out
    finally:
        if proc.poll() is None:
            os.kill(proc.pid, signal.SIGKILL)           # this exact process, nothing else
        proc.wait()
        del rm
        for f in ("rm.npy", "ixs.npy", "started.npy", "seconds.npy"):
            try:
                os.remove(os.path.join(work, f))
            except OSError:
                pass


# --------------------------------------------------------------------------- kNN + pooling
def knn_search(space: np.ndarray, k: int, include_self: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Exact Euclidean kNN (what sklearn NearestNeighbors computes for neighbors.py:363-376,
    239-243, 282 and analysis.py:1547-1549): fp64 brute force, nearest first, ties by index.
    include_self=False is ``kneighbors_graph(X=None)`` (query excluded); include_self=True is
    ``kneighbors(data)`` where the query point itself comes back in column 0."""
    X = _c64(space)
    n = X.shape[0]
    kk = k if include_self else k + 1
    idx = np.empty((n, k), dtype=np.int64)
    dist = np.empty((n, k))
    blk = max(1, min(n, int(4e7 // max(n * X.shape[1], 1))))
    for s in range(0, n, blk):
        q = X[s:s + blk]
        d2 = ((q[:, None, :] - X[None, :, :]) ** 2).sum(-1)
        if not include_self:
            d2[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = -1.0  # force self first, then drop it
        order = np.lexsort((np.broadcast_to(np.arange(n), d2.shape), d2), axis=1)[:, :kk]
        if not include_self:
            order = order[:, 1:]
        idx[s:s + blk] = order
        diff = q[:, None, :] - X[order]
        dist[s:s + blk] = np.sqrt((diff * diff).sum(-1))
    return dist, idx


def knn_graph(space, k, mode="distance") -> sparse.csr_matrix:
    """knn_distance_matrix (neighbors.py:363-376): CSR (C,C), k entries per row, nearest first."""
    dist, idx = knn_search(space, k, include_self=False)
    n = idx.shape[0]
    data = dist.ravel() if mode == "distance" else np.ones(n * k)
    return sparse.csr_matrix((data, idx.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))


def knn_balance(dsi, dist=None, maxl=200, k=60, constraint=None):
    """knn_balance + balance_knn_loop[_constrained] (neighbors.py:143-183, 11-140)."""
    dsi = np.ascontiguousarray(dsi, dtype=np.int64)
    n, K = dsi.shape
    assert K >= k, "sight needs to be bigger than k"
    l0 = np.bincount(dsi.ravel(), minlength=n)
    lsi = np.ascontiguousarray(np.argsort(l0, kind="mergesort")[::-1], dtype=np.int64)
    return_distance = dist is not None
    if dist is None:
        dist = np.ones(dsi.shape)
        dist[:, 0] = 0
    dist = _c64(dist)
    groups = None if constraint is None else np.ascontiguousarray(constraint, dtype=np.int64)
    dist_new = np.empty((n, k + 1))
    dsi_new = np.empty((n, k + 1), dtype=np.int64)
    l = np.empty(n, dtype=np.int64)
    lib().vo_balance_knn(_ip(dsi), _dp(dist), _ip(lsi), _ip(groups) if groups is not None else None,
                         n, K, int(maxl), int(k), int(return_distance), _dp(dist_new), _ip(dsi_new), _ip(l))
    return dist_new, dsi_new, l


def balanced_knn_graph(space, k, sight_k, maxl, constraint=None):
    """BalancedKNN.fit + kneighbors + kneighbors_graph (neighbors.py:226-322)."""
    dist, dsi = knn_search(space, sight_k + 1, include_self=True)
    dist_new, dsi_new, l = knn_balance(dsi, dist, maxl=maxl, k=k, constraint=constraint)
    n = dsi.shape[0]
    bknn = sparse.csr_matrix((dist_new.ravel(), dsi_new.ravel(), np.arange(0, n * (k + 1) + 1, k + 1)), shape=(n, n))
    return bknn, dist_new, dsi_new, l


def connectivity_to_weights(knn: sparse.spmatrix, diag: float = 1.0) -> sparse.csr_matrix:
    """analysis.py:1006-1010 + neighbors.py:385-390: (knn > 0), diagonal := diag, rows scaled to sum 1."""
    conn = (sparse.csr_matrix(knn) > 0).astype(float).tolil()
    conn.setdiag(diag)
    conn = conn.tocsr()
    rs = np.asarray(conn.sum(1)).ravel()
    return sparse.diags(1.0 / rs) @ conn


def convolve_by_sparse_weights(data, w: sparse.spmatrix, threads=0) -> np.ndarray:
    """neighbors.py:416-423: data (G,C) @ w.T, columns of w.T must sum to one."""
    w = sparse.csr_matrix(w)
    w.sort_indices()
    assert np.allclose(np.asarray(w.sum(1)).ravel(), 1), "weight matrix need to sum to one over the columns"
    D = _c64(data)
    out = np.empty_like(D)
    indptr = np.ascontiguousarray(w.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(w.indices, dtype=np.int64)
    vals = _c64(w.data)
    lib().vo_convolve_csr(_dp(D), _ip(indptr), _ip(indices), _dp(vals), _dp(out), D.shape[0], D.shape[1], int(threads))
    return out


def knn_imputation(S_sz, U_sz, space, k, diag=1.0, maximum=False, balanced=False, b_sight=None, b_maxl=None,
                   constraint=None):
    """VelocytoLoom.knn_imputation (analysis.py:982-1023).  Returns knn, w, Sx, Ux."""
    N = S_sz.shape[1]
    if balanced:
        if b_sight is None:
            b_sight = int(np.maximum(int(k * 8), N - 1))
        if b_maxl is None:
            b_maxl = int(np.maximum(int(k * 4), N - 1))
        knn = balanced_knn_graph(space, k, b_sight, b_maxl, constraint)[0]
    else:
        knn = knn_graph(space, k, mode="distance")
    w = connectivity_to_weights(knn, diag)
    Sx = convolve_by_sparse_weights(S_sz, w)
    Ux = convolve_by_sparse_weights(U_sz, w)
    if maximum:
        Sx = np.maximum(S_sz, Sx)
        Ux = np.maximum(U_sz, Ux)
    return knn, w, Sx, Ux


# --------------------------------------------------------------------------- normalisation
def normalize_size(M, relative_size=None, target_size=None, fix_nonfinite=False):
    """_normalize_S / _normalize_U (analysis.py:535-582): M * (avg_size / cell_size)."""
    M = _c64(M)
    cell_size = M.sum(0) if relative_size is None else np.asarray(relative_size, dtype=float)
    avg = cell_size.mean() if target_size is None else target_size
    with np.errstate(divide="ignore", invalid="ignore"):
        out = (avg / cell_size) * M
    if fix_nonfinite:
        out[~np.isfinite(out)] = 0
    return out, cell_size


# --------------------------------------------------------------------------- gamma fits
def fit_slope(Y, X) -> np.ndarray:
    """estimation.fit_slope (estimation.py:267-279)."""
    Yc, Xc = _c64(Y), _c64(X)
    out = np.empty(Yc.shape[0], dtype=np.float32)
    lib().vo_fit_slope(_dp(Yc), _dp(Xc), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), Yc.shape[0], Yc.shape[1])
    return out


def _up_gamma(y, x, limit_gamma):
    """estimation.py:199-205 / 228-236."""
    if not limit_gamma:
        return 20.0
    if np.median(y) > np.median(x):
        high_x = x > np.percentile(x, 90)
        return float(np.maximum(1.5, np.percentile(y[high_x], 10) / np.median(x[high_x])))
    return 1.5


def _r2(m, q, x, y):
    """estimation.py:323-331 / 355-363 (unweighted coefficient of determination, -1e16 if non-finite)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        r2 = 1 - np.sum((m * x + q - y) ** 2) / np.sum((y.mean() - y) ** 2)
    return r2 if np.isfinite(r2) else -1e16


def box_wls2(x, y, w, lo_m, hi_m, lo_q, hi_q) -> Tuple[float, float]:
    """Exact minimiser of sum w (x m + q - y)^2 over the box [lo_m,hi_m] x [lo_q,hi_q]: a convex
    quadratic in two variables -> interior stationary point if feasible, else the best of the four
    clipped edge minima.  This is the well-posed problem estimation.py:237-240 hands to L-BFGS-B."""
    sw, sx, sy = np.sum(w), np.sum(w * x), np.sum(w * y)
    sxx, sxy = np.sum(w * x * x), np.sum(w * x * y)
    return box_wls2_moments(sw, sx, sy, sxx, sxy, lo_m, hi_m, lo_q, hi_q)


def box_wls2_moments(sw, sx, sy, sxx, sxy, lo_m, hi_m, lo_q, hi_q) -> Tuple[float, float]:
    def f(m, q):  # objective up to the constant sum w y^2
        return m * m * sxx + q * q * sw + 2 * m * q * sx - 2 * m * sxy - 2 * q * sy

    det = sxx * sw - sx * sx
    cands = []
    if det > 0:
        m = (sxy * sw - sx * sy) / det
        q = (sxx * sy - sx * sxy) / det
        if lo_m <= m <= hi_m and lo_q <= q <= hi_q:
            return float(m), float(q)
    for q in (lo_q, hi_q):  # edges q fixed
        m = np.clip((sxy - q * sx) / sxx, lo_m, hi_m) if sxx > 0 else lo_m
        cands.append((f(m, q), m, q))
    for m in (lo_m, hi_m):  # edges m fixed
        q = np.clip((sy - m * sx) / sw, lo_q, hi_q) if sw > 0 else lo_q
        cands.append((f(m, q), m, q))
    _, m, q = min(cands, key=lambda t: t[0])
    return float(m), float(q)


def fit_slope_weighted_offset(Y, X, W, fixperc_q=False, limit_gamma=False, exact=False):
    """estimation.fit_slope_weighted_offset + _fit1_slope_weighted_offset (estimation.py:212-241, 337-366).
    exact=False replays the reference's SciPy calls; exact=True solves the same box-constrained
    problem in closed form (what the GPU path does).  Returns float32 (slopes, offsets, R2)."""
    Y, X, W = _c64(Y), _c64(X), _c64(W)
    G = Y.shape[0]
    slopes = np.zeros(G, dtype=np.float32)
    offsets = np.zeros(G, dtype=np.float32)
    R2 = np.zeros(G, dtype=np.float32)
    for g in range(G):
        y, x, w = Y[g], X[g], W[g]
        if not np.any(x):
            m, q = np.nan, 0.0
        elif not np.any(y):
            m, q = 0.0, 0.0
        elif fixperc_q:
            q = np.percentile(y[x <= np.percentile(x, 1)], 50)
            if exact:
                sxx = np.sum(w * x * x)
                m = float(np.clip(np.sum(w * x * (y - q)) / sxx, 0, 20)) if sxx > 0 else 0.0
            else:
                m = scipy.optimize.minimize_scalar(lambda m: np.sum(w * (x * m - y + q) ** 2), bounds=(0, 20), method="bounded").x
        else:
            up_gamma = _up_gamma(y, x, limit_gamma)
            up_q = 2 * np.sum(y * w) / np.sum(w)
            if exact:
                m, q = box_wls2(x, y, w, 1e-8, up_gamma, 0.0, up_q)
            else:
                m, q = scipy.optimize.minimize(lambda m: np.sum(w * (-y + x * m[0] + m[1]) ** 2), x0=(0.1, 1e-16),
                                               method="L-BFGS-B", bounds=[(1e-8, up_gamma), (0, up_q)]).x
        slopes[g], offsets[g] = m, q
        R2[g] = _r2(m, q, x, y)
    return slopes, offsets, R2


def fit_slope_weighted(Y, X, W, limit_gamma=False, bounds=(0, 20), exact=False):
    """estimation.fit_slope_weighted + _fit1_slope_weighted (estimation.py:191-209, 300-334).
    NB the row loop passes limit_gamma positionally and never forwards `bounds` (:320)."""
    Y, X, W = _c64(Y), _c64(X), _c64(W)
    G = Y.shape[0]
    slopes = np.zeros(G, dtype=np.float32)
    R2 = np.zeros(G, dtype=np.float32)
    for g in range(G):
        y, x, w = Y[g], X[g], W[g]
        if not np.any(x):
            m = np.nan
        elif not np.any(y):
            m = 0.0
        else:
            lo, hi = ((1e-8, _up_gamma(y, x, True)) if limit_gamma else (0, 20))
            if exact:
                sxx = np.sum(w * x * x)
                m = float(np.clip(np.sum(w * x * y) / sxx, lo, hi)) if sxx > 0 else lo
            else:
                m = scipy.optimize.minimize_scalar(lambda m: np.sum(w * (x * m - y) ** 2), bounds=(lo, hi), method="bounded").x
        slopes[g] = m
        R2[g] = _r2(m, 0.0, x, y)
    return slopes, R2


def fit_slope_offset(Y, X, fixperc_q=False, exact=False):
    """estimation.fit_slope_offset + _fit1_slope_offset (estimation.py:244-264, 282-297)."""
    Y, X = _c64(Y), _c64(X)
    G = Y.shape[0]
    slopes = np.zeros(G, dtype=np.float32)
    offsets = np.zeros(G, dtype=np.float32)
    for g in range(G):
        y, x = Y[g], X[g]
        if not np.any(x):
            m, q = np.nan, 0.0
        elif not np.any(y):
            m, q = 0.0, 0.0
        elif fixperc_q:
            q = np.percentile(y[x <= np.percentile(x, 1)], 50)
            if exact:
                m = float(np.clip(np.sum(x * (y - q)) / np.sum(x * x), 0, 20))
            else:
                m = scipy.optimize.minimize_scalar(lambda m: np.sum((x * m - y + q) ** 2), bounds=(0, 20), method="bounded").x
        else:
            if exact:  # ordinary least squares with intercept (what leastsq converges to)
                n = x.size
                det = n * np.sum(x * x) - np.sum(x) ** 2
                m = (n * np.sum(x * y) - np.sum(x) * np.sum(y)) / det
                q = (np.sum(y) - m * np.sum(x)) / n
            else:
                (m, q), _ = scipy.optimize.leastsq(lambda m: -y + x * m[0] + m[1], x0=(0, 0))
        slopes[g], offsets[g] = m, q
    return slopes, offsets


def gamma_weights(Sx, Ux, tmpS, tmpU, weights="maxmin_diag", maxmin_perc=(2, 98), maxmin_weighted_pow=15):
    """The W construction of VelocytoLoom.fit_gammas (analysis.py:1179-1219)."""
    Sx, Ux, tmpS, tmpU = _c64(Sx), _c64(Ux), _c64(tmpS), _c64(tmpU)
    perc = list(maxmin_perc)
    if isinstance(weights, np.ndarray):
        return weights
    if weights == "sum":
        return tmpS / np.percentile(tmpS, 99, 1)[:, None] + tmpU / np.percentile(tmpU, 99, 1)[:, None]
    if weights == "prod":
        return (tmpS / np.percentile(tmpS, 99, 1)[:, None]) * (tmpU / np.percentile(tmpU, 99, 1)[:, None])
    if weights == "maxmin_weighted":
        down, up = np.percentile(tmpS, perc, 1)
        R = np.clip(tmpS, down[:, None], up[:, None])
        R = R - R.min(1)[:, None]
        R = R / R.max(1)[:, None]
        return 0.5 * (R ** maxmin_weighted_pow + (1 - R) ** maxmin_weighted_pow)
    if weights == "maxmin":
        down, up = np.percentile(tmpS, perc, 1)
        return ((tmpS <= down[:, None]) | (tmpS >= up[:, None])).astype(float)
    if weights in ("maxmin_diag", "maxmin_double"):
        dS = np.percentile(Sx, 99.9, 1)
        z = dS == 0
        if z.any():
            dS[z] = np.maximum(Sx[z].max(1), 0.001)
        dU = np.percentile(Ux, 99.9, 1)
        z = dU == 0
        if z.any():
            dU[z] = np.maximum(Ux[z].max(1), 0.001)
        Xn = Sx / dS[:, None] + Ux / dU[:, None]
        down, up = np.percentile(Xn, perc, axis=1)
        W = ((Xn <= down[:, None]) | (Xn >= up[:, None])).astype(float)
        if weights == "maxmin_double":
            down, up = np.percentile(Sx, perc, 1)
            W = W + ((Sx <= down[:, None]) | (Sx >= up[:, None])).astype(float)
        return W
    raise ValueError(weights)


def fit_gammas(Sx, Ux, Sx_sz, Ux_sz, fit_offset=True, fixperc_q=False, weighted=True, weights="maxmin_diag",
               limit_gamma=False, maxmin_perc=(2, 98), maxmin_weighted_pow=15, use_size_norm=True, exact=False, steady_state=None):
    """VelocytoLoom.fit_gammas with use_imputed_data=True (analysis.py:1120-1260).  Returns gammas, q, R2.
    `steady_state`: boolean mask over cells; the data are subset as at analysis.py:1223-1257 and - what the reference forgets,
    so that its weighted fits fail on broadcasting for any mask that drops a cell - the weights W (computed over ALL cells,
    :1179-1219) are restricted to the same cells."""
    tmpS, tmpU = (Sx_sz, Ux_sz) if use_size_norm else (Sx, Ux)
    R2 = None
    if weighted:
        W = gamma_weights(Sx, Ux, tmpS, tmpU, weights, maxmin_perc, maxmin_weighted_pow)
    if steady_state is not None:
        ss = np.asarray(steady_state, dtype=bool)
        tmpS, tmpU = tmpS[:, ss], tmpU[:, ss]
        if weighted:
            W = W[:, ss]
    if fit_offset:
        if weighted:
            g, q, R2 = fit_slope_weighted_offset(tmpU, tmpS, W, limit_gamma=limit_gamma, exact=exact)
        else:
            g, q = fit_slope_offset(tmpU, tmpS, exact=exact)
    elif fixperc_q:
        if weighted:
            g, q, _ = fit_slope_weighted_offset(tmpU, tmpS, W, fixperc_q=True, limit_gamma=limit_gamma, exact=exact)
        else:
            g, q = fit_slope_offset(tmpU, tmpS, fixperc_q=True, exact=exact)
    else:
        if weighted:
            g, R2 = fit_slope_weighted(tmpU, tmpS, W, limit_gamma=limit_gamma, exact=exact)
        else:
            g = fit_slope(tmpU, tmpS)
        q = np.zeros_like(g)
    g = g.copy()
    g[~np.isfinite(g)] = 0
    return g, q, R2


# --------------------------------------------------------------------------- velocity / extrapolation
def velocity_chain(Sx_sz, Ux_sz, gammas, q, delta_t_shift=1.0, delta_t_extrap=1.0, assumption="constant_velocity",
                   eps=None, clip=True):
    """predict_U -> calculate_velocity -> calculate_shift -> extrapolate_cell_at_t
    (analysis.py:1321-1439).  Returns Upred, velocity, delta_S, Sx_sz_t."""
    Sx_sz, Ux_sz = _c64(Sx_sz), _c64(Ux_sz)
    g = np.asarray(gammas)[:, None]  # float32 in the reference; numpy promotes the product to fp64
    qq = np.zeros_like(g) if q is None else np.asarray(q)[:, None]
    Upred = g * Sx_sz + qq
    velocity = Ux_sz - Upred
    if eps:
        thr = Upred.max(1) * eps
        velocity[np.abs(velocity) < thr[:, None]] = 0
    if assumption == "constant_velocity":
        delta_S = delta_t_shift * velocity
    elif assumption == "constant_unspliced":
        Uo = Ux_sz - qq
        Uo[Uo < 0] = 0
        with np.errstate(divide="ignore", invalid="ignore"):
            egt = np.exp(-np.asarray(gammas) * delta_t_shift)[:, None]
            delta_S = Sx_sz * egt + (1 - egt) * Uo / np.asarray(gammas)[:, None] - Sx_sz
    else:
        raise NotImplementedError(assumption)
    Sx_sz_t = Sx_sz + delta_t_extrap * delta_S
    if clip:
        Sx_sz_t = np.clip(Sx_sz_t, 0, None)
    return Upred, velocity, delta_S, Sx_sz_t


def delta_transform(hi_dim, hi_dim_t, transform, psc):
    """The `dmat` argument built at analysis.py:1575-1601 (knn_random) / :1637-1663 (full)."""
    if transform == "log":
        d = hi_dim_t - hi_dim
        return np.log10(np.abs(d) + psc) * np.sign(d)
    if transform == "sqrt":
        d = hi_dim_t - hi_dim
        return np.sqrt(np.abs(d) + psc) * np.sign(d)
    if transform == "linear":
        return hi_dim_t - hi_dim
    if transform == "logratio":
        return np.log2(np.abs(hi_dim_t) + psc) - np.log2(hi_dim + psc)
    raise NotImplementedError(transform)


def default_psc(transform, psc=None):
    """analysis.py:1520-1526."""
    if psc is not None:
        return psc
    return 1.0 if transform in ("log", "logratio") else (1e-10 if transform == "sqrt" else 0)


def sample_neighbors(embedding, n_neighbors, sampled_fraction=0.3, sampling_probs=(0.5, 0.1), random_seed=15071990):
    """Embedding kNN + per-cell weighted subsampling (analysis.py:1529, 1547-1572), legacy numpy RNG stream.
    Returns (neigh_ixs (C, nrndm), sampling_ixs, full_knn_ixs (C, n_neighbors+1))."""
    np.random.seed(random_seed)
    _, knn_ixs = knn_search(embedding, n_neighbors + 1, include_self=False)
    p = np.linspace(sampling_probs[0], sampling_probs[1], knn_ixs.shape[1])
    p = p / p.sum()
    size = int(sampled_fraction * (n_neighbors + 1))
    sampling_ixs = np.stack([np.random.choice(knn_ixs.shape[1], size=(size,), replace=False, p=p)
                             for _ in range(knn_ixs.shape[0])], 0)
    neigh_ixs = knn_ixs[np.arange(knn_ixs.shape[0])[:, None], sampling_ixs]
    return neigh_ixs, sampling_ixs, knn_ixs


def estimate_transition_prob(hi_dim, delta_S, embedding, used_delta_t=1.0, transform="sqrt", psc=None,
                             n_neighbors=None, knn_random=True, sampled_fraction=0.3, sampling_probs=(0.5, 0.1),
                             random_seed=15071990, neigh_ixs=None, threads=0):
    """VelocytoLoom.estimate_transition_prob without the randomised control (analysis.py:1452-1668).
    Returns corrcoef (C,C) dense, neigh_ixs (knn_random) or the full embedding kNN indices."""
    hi_dim, delta_S = _c64(hi_dim), _c64(delta_S)
    C = hi_dim.shape[1]
    if n_neighbors is None:
        n_neighbors = int(C / 5)
    psc = default_psc(transform, psc)
    hi_dim_t = hi_dim + used_delta_t * delta_S
    if transform == "logratio":
        e = np.log2(hi_dim + psc)
        d = np.log2(np.abs(hi_dim_t) + psc) - e
        kern = "linear"
    else:
        e = hi_dim
        d = delta_transform(hi_dim, hi_dim_t, transform, psc)
        kern = {"log": "log10", "sqrt": "sqrt", "linear": "linear"}[transform]
    if knn_random:
        if neigh_ixs is None:
            neigh_ixs = sample_neighbors(embedding, n_neighbors, sampled_fraction, sampling_probs, random_seed)[0]
        cc = coldeltacor_partial(e, d, neigh_ixs, kern, psc, threads)
        np.fill_diagonal(cc, 0)
        cc[np.isnan(cc)] = 1
        return cc, neigh_ixs
    cc = coldeltacor(e, d, kern, psc, threads)
    np.fill_diagonal(cc, 0)
    _, knn_ixs = knn_search(embedding, n_neighbors + 1, include_self=False)
    return cc, knn_ixs


def calculate_embedding_shift(corrcoef, neigh_ixs, embedding, hi_dim=None, delta_S=None, sigma_corr=0.05,
                              expression_scaling=True, scaling_penalty=1.0):
    """VelocytoLoom.calculate_embedding_shift (analysis.py:1670-1733), dense like the reference.
    `neigh_ixs` (C, n) lists the non-zeros of embedding_knn row by row.
    Returns transition_prob (C,C), delta_embedding (C,2), scaling (C,) or None."""
    C = corrcoef.shape[0]
    knn = np.zeros((C, C))
    np.add.at(knn, (np.repeat(np.arange(C), neigh_ixs.shape[1]), neigh_ixs.ravel()), 1.0)
    emb = _c64(embedding)
    tp = np.exp(corrcoef / sigma_corr) * knn
    tp /= tp.sum(1)[:, None]
    unit = emb.T[:, None, :] - emb.T[:, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        unit /= np.linalg.norm(unit, ord=2, axis=0)
        for a in range(unit.shape[0]):
            np.fill_diagonal(unit[a], 0)
    de = (tp * unit).sum(2)
    de -= (knn * unit).sum(2) / knn.sum(1)[None, :]
    de = de.T
    scaling = None
    if expression_scaling:
        estim = hi_dim.dot(tp.T) - hi_dim.dot((knn / knn.sum(1)[:, None]).T)
        with np.errstate(divide="ignore", invalid="ignore"):
            cos_proj = (delta_S * estim).sum(0) / np.sqrt((estim ** 2).sum(0))
        scaling = np.clip(cos_proj / scaling_penalty, 0, 1)
        de = de * scaling[:, None]
    return tp, de, scaling


def gaussian_kernel(X, mu=0.0, sigma=1.0):
    """analysis.py:2449-2451."""
    return np.exp(-(X - mu) ** 2 / (2 * sigma ** 2)) / np.sqrt(2 * np.pi * sigma ** 2)


def prepare_markov(transition_prob, embedding, sigma_D, sigma_W, direction="forward", cells_ixs=None):
    """VelocytoLoom.prepare_markov (analysis.py:1818-1863); dense result over the cells of cells_ixs (default: all)."""
    if cells_ixs is None:
        cells_ixs = np.arange(np.asarray(transition_prob).shape[0])
    sub = np.asarray(transition_prob)[cells_ixs, :][:, cells_ixs]
    tr = np.array(sub) if direction == "forward" else np.array(sub.T, order="C")
    emb = _c64(embedding)[cells_ixs, :]
    dist = np.sqrt(((emb[:, None, :] - emb[None, :, :]) ** 2).sum(-1))
    tr = tr * gaussian_kernel(dist, sigma=sigma_D)
    np.fill_diagonal(tr, tr.max(1))
    tr = tr / tr.sum(1)[:, None]
    K_W = gaussian_kernel(dist, sigma=sigma_W)
    K_W = K_W / K_W.sum(1)[:, None]
    tr = 0.8 * tr + 0.2 * K_W
    return tr / tr.sum(1)[:, None]


def diffuse(x, tr, n_steps=10, mode="path_integral"):
    """Diffusion.diffuse, modes path_integral / time_evolution (diffusion.py:93-105)."""
    v = np.asarray(x, dtype=np.float64)
    v = (v / v.sum())[None, :]
    T = np.asarray(tr.todense()) if sparse.issparse(tr) else np.asarray(tr)
    acc = np.zeros(np.asarray(x).shape)
    for _ in range(n_steps):
        v = v @ T
        if mode == "path_integral":
            acc = acc + v
    return acc if mode == "path_integral" else v


# --------------------------------------------------------------------------- "next" rows
def knn_query(points, queries, k) -> Tuple[np.ndarray, np.ndarray]:
    """NearestNeighbors(...).fit(points).kneighbors(queries): exact fp64 brute force, nearest first, ties by index."""
    X, Qm = _c64(points), _c64(queries)
    d2 = ((Qm[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    order = np.lexsort((np.broadcast_to(np.arange(X.shape[0]), d2.shape), d2), axis=1)[:, :k]
    return np.sqrt(np.take_along_axis(d2, order, 1)), order


def _norm_pdf(x, scale):
    return np.exp(-0.5 * (x / scale) ** 2) / (scale * np.sqrt(2 * np.pi))


def calculate_grid_arrows(embedding, delta_embedding, smooth=0.5, steps=(40, 40), n_neighbors=100):
    """VelocytoLoom.calculate_grid_arrows (analysis.py:1735-1816).  Returns flow_grid, flow, flow_norm,
    flow_norm_magnitude, total_p_mass."""
    emb = _c64(embedding)
    grs = []
    for i in range(emb.shape[1]):
        m, M = np.min(emb[:, i]), np.max(emb[:, i])
        m = m - 0.025 * np.abs(M - m)
        M = M + 0.025 * np.abs(M - m)            # (sic: uses the already widened m, analysis.py:1785-1786)
        grs.append(np.linspace(m, M, steps[i]))
    grid = np.vstack([g.flat for g in np.meshgrid(*grs)]).T
    dists, neighs = knn_query(emb, grid, n_neighbors)
    std = np.mean([g[1] - g[0] for g in grs])
    w = _norm_pdf(dists, smooth * std)
    mass = w.sum(1)
    UZ = (np.asarray(delta_embedding)[neighs] * w[:, :, None]).sum(1) / np.maximum(1, mass)[:, None]
    mag = np.linalg.norm(UZ, axis=1)
    flow_norm = UZ / np.percentile(mag, 99.5)
    return grid, UZ, flow_norm, np.linalg.norm(flow_norm, axis=1), mass


def _l1_rows(M):
    s = np.abs(M).sum(1)
    s[s == 0] = 1.0
    return M / s[:, None]


def compute_transition_matrix2(x0, v, sigma, reverse=False, n_neighbors=20):
    """Diffusion.compute_transition_matrix2 (diffusion.py:14-53), dense."""
    x0, v = _c64(x0), _c64(v)
    x1 = x0 - v if reverse else x0 + v
    dists, nearest = knn_query(x0, x1, n_neighbors)
    n = x0.shape[0]
    tr = np.zeros((n, n))
    np.add.at(tr, (np.repeat(np.arange(n), n_neighbors), nearest.ravel()), _norm_pdf(dists.ravel(), sigma))
    return _l1_rows(tr)


def compute_transition_matrix(knn_row, knn_col, x, v, epsilon=0.0, reverse=False):
    """Diffusion.compute_transition_matrix (diffusion.py:55-91), dense (n, n)."""
    x, v = _c64(x), _c64(v)
    uv = x[knn_col] - x[knn_row]
    norms = np.linalg.norm(uv, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = uv / norms[:, None]
        proj = (v[knn_row] * uv).sum(1)
        if reverse:
            proj = -proj
        proj = np.clip(proj + epsilon, 0, None)
        p = proj * (1 / norms)
    n = x.shape[0]
    tr = np.zeros((n, n))
    np.add.at(tr, (knn_row, knn_col), p)
    return _l1_rows(tr)


def phase_portrait_filter(R2, gammas, Sx_sz, Ux_sz, minR2=0.1, min_gamma=0.01, minCorr=0.1):
    """The gene mask of VelocytoLoom.filter_genes_by_phase_portrait (analysis.py:1267-1308)."""
    keep = np.ones(np.asarray(gammas).shape, dtype=bool)
    if minR2 is not None:
        keep &= (np.sqrt(np.abs(R2)) * np.sign(R2)) > minR2
    if min_gamma is not None:
        keep &= np.asarray(gammas) > min_gamma
    if minCorr is not None:
        A, B = _c64(Sx_sz), _c64(Ux_sz)
        A_m, B_m = A - A.mean(1)[:, None], B - B.mean(1)[:, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            corr = (A_m * B_m).sum(1) / (np.linalg.norm(A_m, 2, 1) * np.linalg.norm(B_m, 2, 1))
        keep &= corr > minCorr
    return keep


# --------------------------------------------------------------------------- callers upstream of the path
def score_detection_levels(S, U, min_expr_counts=50, min_cells_express=20, min_expr_counts_U=0, min_cells_express_U=0):
    """VelocytoLoom.score_detection_levels (analysis.py:456-475)."""
    S, U = np.asarray(S), np.asarray(U)
    return ((S.sum(1) >= min_expr_counts) & ((S > 0).sum(1) >= min_cells_express) &
            (U.sum(1) >= min_expr_counts_U) & ((U > 0).sum(1) >= min_cells_express_U))


def score_cv_vs_mean(S, N=3000, min_expr_cells=2, max_expr_avg=20, min_expr_avg=0, svr_gamma=None, winsorize=False,
                     winsor_perc=(1, 99.5), sort_inverse=False):
    """VelocytoLoom.score_cv_vs_mean for one layer (analysis.py:201-345; SVR = scikit-learn, as in the reference).
    Returns (score over all genes, selected mask)."""
    from sklearn.svm import SVR
    S = _c64(S)
    if winsorize and min_expr_cells <= ((100 - winsor_perc[1]) * S.shape[1] * 0.01):
        min_expr_cells = int(np.ceil((100 - winsor_perc[1]) * S.shape[0] * 0.01)) + 2          # :248 (shape[0], sic)
    detected = ((S > 0).sum(1) > min_expr_cells) & (S.mean(1) < max_expr_avg) & (S.mean(1) > min_expr_avg)
    Sf = S[detected]
    if winsorize:
        down, up = np.percentile(Sf, winsor_perc, 1)
        Sf = np.clip(Sf, down[:, None], up[:, None])
    mu, sigma = Sf.mean(1), Sf.std(1, ddof=1)
    log_m, log_cv = np.log2(mu), np.log2(sigma / mu)
    clf = SVR(gamma=150. / len(mu) if svr_gamma is None else svr_gamma)
    clf.fit(log_m[:, None], log_cv)
    score = log_cv - clf.predict(log_m[:, None])
    if sort_inverse:
        score = -score
    nth = np.sort(score)[::-1][N]
    full = np.zeros(detected.shape)
    full[~detected] = np.min(score) - 1e-16
    full[detected] = score
    return full, full >= nth


def clusters_stats(U, S, cluster_ix, n_clusters, size_limit=40):
    """estimation.clusters_stats (estimation.py:369-389): per-cluster gene means; small clusters get the overall mean."""
    U, S = _c64(U), _c64(S)
    U_avgs, S_avgs = np.zeros((S.shape[0], n_clusters)), np.zeros((S.shape[0], n_clusters))
    for i in range(n_clusters):
        f = cluster_ix == i
        if f.sum() > size_limit:
            U_avgs[:, i], S_avgs[:, i] = U[:, f].mean(1), S[:, f].mean(1)
        else:
            U_avgs[:, i], S_avgs[:, i] = U.mean(1), S.mean(1)
    return U_avgs, S_avgs


def robust_size_factor(S, selected, pc=0.1):
    """VelocytoLoom.robust_size_factor for one layer (analysis.py:347-439)."""
    Y = np.log2(_c64(S)[selected] + pc)
    sf = np.median(2**(Y - Y.mean(1)[:, None]), axis=0)
    return sf / np.mean(sf)


def normalize_by_total(S, U, initial_cell_size, initial_Ucell_size, min_perc_U=0.5, skip_low_U_pop=True, same_size_UnS=False,
                       size_factor=None):
    """normalize_by_total (analysis.py:704-758), or normalize_by_size_factor (:760-818) when size_factor is given (the cell
    sizes then come from the current S / U).  Returns (S_sz, U_sz, small_U_pop)."""
    S, U = _c64(S), _c64(U)
    cs, ucs = (initial_cell_size, initial_Ucell_size) if size_factor is None else (S.sum(0), U.sum(0))
    target = np.median(cs)
    min_U = np.percentile(ucs, min_perc_U)
    if min_U < 2:
        raise ValueError("min_perc_U corresponds to total Unspliced of 1 molecule or less")
    small = ucs < min_U
    target_U = target if same_size_UnS else np.median(ucs[~small])
    S_sz = normalize_size(S, initial_cell_size if size_factor is None else size_factor, target)[0]
    rel_U = np.clip(initial_Ucell_size, min_U, None) if skip_low_U_pop else initial_Ucell_size
    U_sz = normalize_size(U, rel_U, target_U, fix_nonfinite=True)[0]
    return S_sz, U_sz, small


def normalize_median_renorm(S_sz, U_sz, small_U_pop, skip_low_U_pop=True):
    """normalize_median(which="renormalize") (analysis.py:876-882)."""
    S_sz, U_sz = _c64(S_sz).copy(), _c64(U_sz).copy()
    S_sz = S_sz * (np.median(S_sz.sum(0)) / S_sz.sum(0))
    m = ~small_U_pop if skip_low_U_pop else np.ones(U_sz.shape[1], dtype=bool)
    U_sz[:, m] = U_sz[:, m] * (np.median(U_sz[:, m].sum(0)) / U_sz[:, m].sum(0))
    return S_sz, U_sz


def adjust_totS_totU(S_sz, U_sz, small_U_pop, skip_low_U_pop=True, normalize_total=False, fit_with_low_U=True, svr_C=100, svr_gamma=1e-6):
    """adjust_totS_totU (analysis.py:820-868)."""
    from sklearn.svm import SVR
    S_sz, U_sz = _c64(S_sz).copy(), _c64(U_sz).copy()
    svr = SVR(C=svr_C, kernel="rbf", gamma=svr_gamma)
    X, y = S_sz.sum(0), U_sz.sum(0)
    if fit_with_low_U:
        svr.fit(X[:, None], y)
        predicted = svr.predict(X[:, None])
    else:
        svr.fit(X[~small_U_pop, None], y[~small_U_pop])
        predicted = y.copy()
        predicted[~small_U_pop] = svr.predict(X[~small_U_pop, None])
    with np.errstate(divide="ignore", invalid="ignore"):
        adj = predicted / y
    adj[~np.isfinite(adj)] = 1
    if skip_low_U_pop:
        U_sz[:, ~small_U_pop] = U_sz[:, ~small_U_pop] * adj[~small_U_pop]
    else:
        U_sz = U_sz * adj
    if normalize_total:
        S_sz, U_sz = normalize_median_renorm(S_sz, U_sz, small_U_pop, skip_low_U_pop)
    return S_sz, U_sz


def normalize_median_imputed(Sx, Ux, small_U_pop, skip_low_U_pop=True):
    """normalize_median(which="imputed") (analysis.py:883-889)."""
    Sx, Ux = _c64(Sx), _c64(Ux)
    Sx_sz = Sx * (np.median(Sx.sum(0)) / Sx.sum(0))
    Ux_sz = Ux.copy()
    m = ~small_U_pop if skip_low_U_pop else np.ones(Ux.shape[1], dtype=bool)
    Ux_sz[:, m] = Ux[:, m] * (np.median(Ux[:, m].sum(0)) / Ux[:, m].sum(0))
    return Sx_sz, Ux_sz


def pca(X, n_components=None):
    """VelocytoLoom.perform_PCA (analysis.py:678-702) = sklearn.decomposition.PCA().fit_transform(X.T) restated: centre the
    genes, SVD, scikit-learn >= 1.5 sign rule (largest-|.| loading of each component positive).  X: (genes, cells).
    Returns (pcs (cells, k), components (k, genes), explained_variance_ratio (k,))."""
    A = _c64(X).T
    A = A - A.mean(0)
    Uu, s, Vt = np.linalg.svd(A, full_matrices=False)
    sign = np.sign(Vt[np.arange(Vt.shape[0]), np.abs(Vt).argmax(1)])
    Uu, Vt = Uu * sign[None, :], Vt * sign[:, None]
    k = min(A.shape) if n_components is None else n_components
    var = s**2 / (A.shape[0] - 1)
    return (Uu * s)[:, :k], Vt[:k], (var / var.sum())[:k]


# --------------------------------------------------------------------------- libsvm's epsilon-SVR (third party, restated)
def svr_rbf_fit(x, t, C=1.0, epsilon=0.1, gamma=1.0, tol=1e-3, max_iter=10_000_000):
    """What ``sklearn.svm.SVR(kernel="rbf", C, epsilon, gamma, tol).fit(x[:, None], t)`` hands to libsvm, restated from the
    published algorithm (Chang & Lin, "LIBSVM", sect. 4; Fan, Chen & Lin, JMLR 2005): the dual over b = [alpha; alpha*],
    y = [+1; -1], p = [eps - t; eps + t], SMO with working-set selection "WSS 2" (i = argmax over I_up of -y G, j = argmin over
    I_low of -(b_ij)^2 / a_ij), the two-variable update with libsvm's clipping order, stop at m(b) - M(b) < tol, rho from the free
    variables.  Ties go to the lowest variable index.  fp64 kernel rows (libsvm rounds them to float and shrinks: same optimum
    within tol).  Returns (coef = alpha - alpha*, intercept = -rho, SMO steps)."""
    x, t = np.asarray(x, dtype=np.float64).ravel(), np.asarray(t, dtype=np.float64).ravel()
    n = len(x)
    y = np.concatenate([np.ones(n), -np.ones(n)])
    beta = np.zeros(2 * n)
    r = t.copy()                                          # residual t - f(x): -y G = r - eps (alpha part), r + eps (alpha* part)
    tau = 1e-12
    it = 0
    while True:
        mG = np.concatenate([r - epsilon, r + epsilon])
        up = ((y > 0) & (beta < C)) | ((y < 0) & (beta > 0))
        low = ((y > 0) & (beta > 0)) | ((y < 0) & (beta < C))
        vi = np.where(up, mG, -np.inf)
        i = int(np.argmax(vi))                            # first maximum = lowest index
        Gmax = vi[i]
        Gmin = np.min(np.where(low, mG, np.inf)) if low.any() else np.inf
        if not up.any() or not (Gmax - Gmin >= tol) or it >= max_iter:
            break
        Ki = np.exp(-gamma * (x - x[i % n]) ** 2)
        a = 2.0 - 2.0 * np.concatenate([Ki, Ki])
        a = np.where(a > 0, a, tau)
        b = Gmax - mG
        cand = low & (b > 0)
        if not cand.any():
            break
        obj = np.where(cand, b * b * (-1.0 / a), np.inf)
        j = int(np.argmin(obj))
        pi_, pj = i % n, j % n
        Gi = epsilon - r[pi_] if i < n else r[pi_] + epsilon
        Gj = epsilon - r[pj] if j < n else r[pj] + epsilon
        Kij = np.exp(-gamma * (x[pi_] - x[pj]) ** 2)
        quad = 2.0 - 2.0 * Kij
        if not quad > 0:
            quad = tau
        ai, aj = beta[i], beta[j]
        if (i < n) != (j < n):
            delta, diff = (-Gi - Gj) / quad, ai - aj
            ai, aj = ai + delta, aj + delta
            if diff > 0:
                if aj < 0: aj, ai = 0.0, diff
            else:
                if ai < 0: ai, aj = 0.0, -diff
            if diff > 0:
                if ai > C: ai, aj = C, C - diff
            else:
                if aj > C: aj, ai = C, C + diff
        else:
            delta, sm = (Gi - Gj) / quad, ai + aj
            ai, aj = ai - delta, aj + delta
            if sm > C:
                if ai > C: ai, aj = C, sm - C
            else:
                if aj < 0: aj, ai = 0.0, sm
            if sm > C:
                if aj > C: aj, ai = C, sm - C
            else:
                if ai < 0: ai, aj = 0.0, sm
        dci = (1.0 if i < n else -1.0) * (ai - beta[i])
        dcj = (1.0 if j < n else -1.0) * (aj - beta[j])
        beta[i], beta[j] = ai, aj
        r -= dci * Ki + dcj * np.exp(-gamma * (x - x[pj]) ** 2)
        it += 1
    yG = np.concatenate([epsilon - r, -(r + epsilon)])
    free = (beta > 0) & (beta < C)
    if free.any():
        rho = yG[free].sum() / free.sum()
    else:
        ub = np.min(np.where(((y > 0) & (beta <= 0)) | ((y < 0) & (beta >= C)), yG, np.inf))
        lb = np.max(np.where(((y > 0) & (beta >= C)) | ((y < 0) & (beta <= 0)), yG, -np.inf))
        rho = 0.5 * (ub + lb)
    return beta[:n] - beta[n:], -rho, it


def svr_rbf_predict(x, coef, intercept, xq, gamma):
    """SVR.predict: sum_k coef_k exp(-gamma (xq - x_k)^2) + intercept."""
    x, xq = np.asarray(x, dtype=np.float64).ravel(), np.asarray(xq, dtype=np.float64).ravel()
    return (coef[None, :] * np.exp(-gamma * (xq[:, None] - x[None, :]) ** 2)).sum(1) + intercept
