"""ctypes binding of libeasyrag_hip.so (include/easyrag_hip.h).  No compute happens in Python.

Loading never falls back to anything: if the library is missing it is built with hipcc
(`easyrag_amd._build`), and if that fails, or no gfx950 device is present when a handle is
created, the caller gets an exception.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from . import _build

_lib = None

ERH_OK = 0
ERH_ERR_INVALID = -1
ERH_ERR_NO_DEVICE = -2
ERH_ERR_HIP = -3
ERH_ERR_STATE = -4
ERH_ERR_UNSUPPORTED = -5
ERH_ERR_OVERFLOW = -6
ERH_ERR_NOMEM = -7

ERH_F16, ERH_F32 = 0, 1
ERH_BM25_OKAPI, ERH_BM25_BM25S = 0, 1
ERH_DENSE_EXACT, ERH_DENSE_FAST = 0, 1
ERH_BM25_SLOTS = 4
ERH_K_DENSE_SCAN, ERH_K_DENSE_SELECT, ERH_K_BM25_SCAN, ERH_K_BM25_MERGE, ERH_K_FUSE, ERH_K_DENSE_SAMPLE = range(6)

_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double

# name -> (restype, argtypes); every symbol include/easyrag_hip.h declares
SIGNATURES = {
    "erh_version": (_i32, []),
    "erh_status_str": (C.c_char_p, [_i32]),
    "erh_create": (_i32, [_i32, C.POINTER(_vp)]),
    "erh_destroy": (_i32, [_vp]),
    "erh_last_error": (C.c_char_p, [_vp]),
    "erh_sync": (_i32, [_vp, _vp]),
    "erh_set_dense": (_i32, [_vp, _vp, _i64, _i32, _i32, _i32, _i32]),
    "erh_get_dense_rows": (_i32, [_vp, _i64, _i64, _vp, _i32]),
    "erh_set_bm25_csr": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp]),
    "erh_set_bm25_tf": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl]),
    "erh_build_bm25_index": (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _vp, _i32, _dbl, _dbl, _dbl, C.POINTER(_i64)]),
    "erh_get_bm25_csr": (_i32, [_vp, _vp, _vp, _vp, _vp, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "erh_bm25_select": (_i32, [_vp, _i32]),
    "erh_bm25_release": (_i32, [_vp, _i32]),
    "erh_get_bm25_payload": (_i32, [_vp, _vp]),
    "erh_set_doc_meta": (_i32, [_vp, _i64, _vp, _vp]),
    "erh_dense_topk": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "erh_bm25_topk": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    "erh_bm25_scores": (_i32, [_vp, _vp, _i32, _vp]),
    "erh_rrf": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "erh_fusion": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "erh_hybrid_topk": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp,
                               _vp, _vp, _vp, _i32, _vp]),
    "erh_comm_unique_id": (_i32, [_vp]),
    "erh_comm_init": (_i32, [_vp, _i32, _i32, _vp]),
    "erh_comm_destroy": (_i32, [_vp]),
    "erh_allgather_topk": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "erh_topk_row_bytes": (_i32, [_i32]),
    "erh_pack_topk": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "erh_unpack_topk": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "erh_set_profiling": (_i32, [_vp, _i32]),
    "erh_get_kernel_time": (_i32, [_vp, _i32, C.POINTER(_dbl), C.POINTER(_i64)]),
    "erh_get_kernel_work": (_i32, [_vp, _i32, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "erh_reset_kernel_time": (_i32, [_vp]),
    "erh_set_option": (_i32, [_vp, C.c_char_p, _i64]),
    "erh_dense_check": (_i32, [_vp, _vp]),
    "erh_debug_counters": (_i32, [_vp, _vp]),
    "erh_dense_diag": (_i32, [_vp, C.POINTER(_dbl), C.POINTER(_dbl), C.POINTER(_i32)]),
    "erh_dense_exhaustive_count": (_i32, [_vp, C.POINTER(_i32)]),
    "erh_get_stat": (_i32, [_vp, C.c_char_p, C.POINTER(_i64)]),
    "erh_reset_stats": (_i32, [_vp]),
    "erh_dense_seed_rank": (_i32, [_i32, _i64, _i64]),
    "erh_debug_dense_scores": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "erh_vocab_create": (_i32, [C.POINTER(_vp)]),
    "erh_vocab_destroy": (_i32, [_vp]),
    "erh_vocab_size": (_i64, [_vp]),
    "erh_vocab_encode": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, C.POINTER(_i64)]),
    "erh_vocab_token": (_i32, [_vp, _i32, C.POINTER(_vp), C.POINTER(_i32)]),
    "erh_cutter_create": (_i32, [_vp, _i64, C.POINTER(_vp)]),
    "erh_cutter_destroy": (_i32, [_vp]),
    "erh_cutter_cut": (_i32, [_vp, _vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "erh_cutter_set_hmm": (_i32, [_vp, _vp, _i64]),
    "erh_cutter_has_hmm": (_i32, [_vp]),
    "erh_cutter_cut_mode": (_i32, [_vp, _vp, _i64, _i32, _vp, _i64, C.POINTER(_i64)]),
    "erh_text_encode": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp, C.POINTER(_i64)]),
    "erh_text_encode_mt": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp, C.POINTER(_i64)]),
}


def lib_path() -> Path:
    return _build.LIB_PATH


def load(build_if_missing: bool = True):
    """dlopen libeasyrag_hip.so (building it in-tree first if needed) and set the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so (same SONAME); import it first so one HIP runtime serves
    # both torch's allocations and this library's kernels.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only; the library works without it
        pass
    path = lib_path()
    if build_if_missing:
        _build.build()                   # idempotent (source digest stamp): edited csrc never runs as a stale .so
    elif not path.exists():
        raise FileNotFoundError(f"{path} is missing; run `python -m easyrag_amd._build`")
    lib = C.CDLL(str(path), mode=getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class ErhError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libeasyrag_hip status {status}: {message}")
        self.status = status
