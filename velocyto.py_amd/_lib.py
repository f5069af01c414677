"""Loader / builder of libvelocyto_hip.so (the C-ABI HIP library, include/velocyto_hip.h).

There is NO CPU fallback: if the library is missing or a GPU is absent every compute entry
point raises.  ``build()`` cross-compiles for gfx950 with hipcc (works without a GPU) and
keeps the .so in-tree so that it travels with the repository snapshot.
"""
from __future__ import annotations

import ctypes
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libvelocyto_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

_lib = None

c_i64, c_int, c_dbl, c_vp, c_sz = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
EXPECTED_ABI = 4                  # vcy_abi_version() of the header this table mirrors

# name -> (restype, argtypes); mirrors include/velocyto_hip.h one to one
SIGNATURES = {
    "vcy_last_error": (ctypes.c_char_p, []),
    "vcy_abi_version": (c_int, []),
    "vcy_device_info": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_i64)]),
    "vcy_clock_probe": (c_int, [c_vp, c_i64, c_i64, c_i64, c_vp]),
    "vcy_transpose": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp]),
    "vcy_coldeltacor_partial": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                        c_int, c_int, c_dbl, c_int, c_vp]),
    "vcy_coldeltacor_partial_fused": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                              c_int, c_int, c_dbl, c_dbl, c_dbl, c_int, c_vp]),
    "vcy_coldeltacor_partial_dual": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                             c_int, c_int, c_dbl, c_int, c_vp]),
    "vcy_coldeltacor_partial_fused_dual": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64,
                                                   c_i64, c_i64, c_int, c_int, c_dbl, c_dbl, c_dbl, c_int, c_vp]),
    "vcy_coldeltacor_full": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_dbl,
                                     c_int, c_int, c_vp]),
    "vcy_coldeltacor_full_linear_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "vcy_coldeltacor_full_linear": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp]),
    "vcy_scatter_rows": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_knn_pool": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_vp]),
    "vcy_knn_pool2": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_vp]),
    "vcy_knn_pool_w2": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_knn_pool_counts": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                    c_int, c_i64, c_int, c_int, c_vp]),
    "vcy_csr_slab_genes": (c_i64, []),
    "vcy_csr_slab_ptr": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "vcy_knn_pool_csr": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64,
                                 c_int, c_int, c_int, c_vp]),
    "vcy_knn_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "vcy_knn_row_free": (c_int, [c_i64, c_i64]),
    "vcy_knn_search": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_knn_query": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "vcy_balance_knn_host": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp, c_vp, c_vp]),
    "vcy_balance_knn_host32": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "vcy_fit_workspace_bytes": (c_sz, [c_i64]),
    "vcy_fit_slope": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_fit_slope_moments": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_fit_slope_from_moments": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "vcy_gene_moments": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_abs_stats_workspace_bytes": (c_i64, []),
    "vcy_abs_stats": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_gene_stats_workspace_bytes": (c_i64, [c_i64]),
    "vcy_gene_stats": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_choice_stream_host": (c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "vcy_markov_factored_workspace_bytes": (c_sz, [c_i64]),
    "vcy_prepare_markov_factored": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_dbl, c_dbl, c_int, c_vp]),
    "vcy_markov_cull_boxes_bytes": (c_sz, [c_i64, c_int, c_int]),
    "vcy_markov_cull_boxes": (c_int, [c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "vcy_diffuse_step_factored_culled": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_dbl, c_dbl, c_vp,
                                                 c_i64, c_int, c_int, c_vp]),
    "vcy_diffuse_step_factored": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_dbl, c_vp, c_i64, c_int, c_int, c_vp]),
    "vcy_gram_workspace_bytes": (c_sz, [c_i64, c_i64, c_i64, c_int]),
    "vcy_col_means": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_gram": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_gram_tn": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_gemm_nt": (c_int, [c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_vp]),
    "vcy_svr_workspace_bytes": (c_i64, [c_i64]),
    "vcy_svr_rbf_fit": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_dbl, c_dbl, c_dbl, c_dbl, c_i64, c_vp]),
    "vcy_svr_rbf_predict": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_dbl, c_vp]),
    "vcy_quantile_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "vcy_gene_quantiles": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, ctypes.POINTER(c_dbl), c_int, c_vp, c_vp, c_i64,
                                   c_i64, c_i64, c_int, c_vp]),
    "vcy_gamma_weights": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_dbl, c_int, c_vp]),
    "vcy_prepare_markov": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_dbl, c_dbl, c_int, c_vp]),
    "vcy_row_sums": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_scale_log": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_dbl, c_int, c_int, c_vp]),
    "vcy_delta_transform": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_dbl, c_int, c_dbl, c_int, c_vp]),
    "vcy_permute_rows_nsign_workspace_bytes": (c_sz, [c_i64, c_i64, c_int]),
    "vcy_permute_rows_nsign": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, ctypes.c_uint64, c_int, c_vp]),
    "vcy_corr_fixup": (c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_dbl, c_vp, c_int, c_vp]),
    "vcy_transition_prob": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_dbl, c_int, c_vp]),
    "vcy_row_cosproj": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_embedding_scaling_max_neighbors": (c_int, []),
    "vcy_embedding_scaling": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_diffuse_workspace_bytes": (c_sz, [c_i64]),
    "vcy_diffuse_step_dense": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "vcy_diffuse_step_csc": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "vcy_fit_weighted": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_dbl, c_dbl,
                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_lincomb": (c_int, [c_vp, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_vp]),
    "vcy_velocity_chain": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64,
                                   c_dbl, c_dbl, c_dbl, c_int, c_int, c_int, c_dbl, c_int, c_vp]),
}


def sources():
    return sorted(glob.glob(os.path.join(_CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(_CSRC, "*.h")) + glob.glob(os.path.join(_PKG, "..", "include", "*.h"))
    return any(os.path.getmtime(f) > t for f in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip -> libvelocyto_hip.so (in-tree)."""
    if not force and not _stale():
        return LIB_PATH
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build libvelocyto_hip.so")
    objdir = os.path.join(_CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC, *flags, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, sources()))
    r = subprocess.run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load the library (torch first, so both share one HIP runtime: same soname libamdhip64.so.7)."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (loads torch's libamdhip64 before ours resolves it)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback for the HIP path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.vcy_abi_version() != EXPECTED_ABI:     # same symbols, other argument lists: never call into it
            raise RuntimeError(f"{LIB_PATH} has ABI version {L.vcy_abi_version()}, this binding was written against {EXPECTED_ABI} "
                               "(include/velocyto_hip.h): rebuild it with `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().vcy_last_error().decode("utf-8", "replace")
        exc = NotImplementedError if rc == -3 else (ValueError if rc == -1 else RuntimeError)
        raise exc(f"libvelocyto_hip {what} failed ({rc}): {msg}")
