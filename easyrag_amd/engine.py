"""RetrievalEngine: thin, typed wrapper over one libeasyrag_hip handle (one GPU).

It only marshals: numpy arrays / torch tensors -> pointers, status codes -> exceptions.  All
retrieval arithmetic runs in the HIP kernels; nothing here computes a score.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .index import BM25Index, BM25S, OKAPI


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _ptr(a) -> int:
    if a is None:
        return 0
    if _is_torch(a):
        return int(a.data_ptr())
    return int(a.ctypes.data)


def _np(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def queries_to_csr(queries: Sequence[Sequence[int]]) -> Tuple[np.ndarray, np.ndarray]:
    """List of term-id sequences -> (q_indptr int32[B+1], q_tok int32[sum])."""
    lens = np.fromiter((len(q) for q in queries), dtype=np.int64, count=len(queries))
    indptr = np.zeros(len(queries) + 1, np.int32)
    np.cumsum(lens, out=indptr[1:])
    tok = np.empty(int(indptr[-1]), np.int32)
    for i, q in enumerate(queries):
        tok[indptr[i]:indptr[i + 1]] = q
    return indptr, tok


class RetrievalEngine:
    """One handle = one GPU's replica of the corpus state (chunk matrix, postings, metadata)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.erh_create(int(device), C.byref(h))
        if rc != 0:
            raise _lib.ErhError(rc, self._lib.erh_status_str(rc).decode() +
                                " (libeasyrag_hip needs a gfx950 GPU; there is no CPU fallback)")
        self._h = h
        self.device = int(device)
        self.n_dense = 0
        self.d = 0
        self._bm25_slots: List[Optional[BM25Index]] = [None] * _lib.ERH_BM25_SLOTS
        self._bm25_cur = 0
        self.n_meta = 0
        self.rank, self.world = 0, 1
        self.corpus = None                   # retrievers.py keeps its per-engine node bookkeeping here

    @property
    def bm25(self) -> Optional[BM25Index]:
        """The BM25 index of the selected slot (None while the slot is empty or only reserved)."""
        v = self._bm25_slots[self._bm25_cur]
        return None if v is self._RESERVED else v

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.erh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise _lib.ErhError(rc, self._lib.erh_last_error(self._h).decode())

    # -- corpus state ---------------------------------------------------------------------------
    def set_dense(self, x, normalize: bool = False):
        """x: [N, d] float16 or float32, numpy array or (ROCm) torch tensor."""
        if _is_torch(x):
            import torch
            x = x.contiguous()
            is_dev = 1 if x.is_cuda else 0
            if x.dtype == torch.float16:
                dt = _lib.ERH_F16
            elif x.dtype == torch.float32:
                dt = _lib.ERH_F32
            else:
                raise TypeError("dense matrix must be float16 or float32")
            n, d = int(x.shape[0]), int(x.shape[1])
            if is_dev:
                torch.cuda.current_stream(x.device).synchronize()
        else:
            x = np.ascontiguousarray(x)
            if x.dtype == np.float16:
                dt = _lib.ERH_F16
            elif x.dtype == np.float32:
                dt = _lib.ERH_F32
            else:
                raise TypeError("dense matrix must be float16 or float32")
            is_dev = 0
            n, d = x.shape
        self._check(self._lib.erh_set_dense(self._h, _ptr(x), n, d, dt, is_dev, 1 if normalize else 0))
        self.n_dense, self.d = int(n), int(d)

    def get_dense_rows(self, row0: int, rows: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Rows [row0, row0 + rows) of the stored fp16 matrix in the caller's order (erh_get_dense_rows)."""
        if out is None:
            out = np.empty((int(rows), self.d), np.float16)
        if out.dtype != np.float16 or not out.flags["C_CONTIGUOUS"] or out.shape != (int(rows), self.d):
            raise ValueError("out must be a C-contiguous float16 array of shape (rows, d)")
        self._check(self._lib.erh_get_dense_rows(self._h, int(row0), int(rows), _ptr(out), 0))
        return out

    # A handle holds ERH_BM25_SLOTS independent BM25 indices (content route + know_path route of the reference
    # pipeline share one engine); every BM25 call names its slot, default 0.
    _RESERVED = "reserved"                  # a slot handed out by alloc_bm25_slot that has no index yet

    def alloc_bm25_slot(self) -> int:
        """Reserve a free slot (it counts as taken from here on, index or not; free_bm25_slot gives it back)."""
        for i, v in enumerate(self._bm25_slots):
            if v is None:
                self._bm25_slots[i] = self._RESERVED
                return i
        hint = (" (one of them is the scratch slot of BM25Retriever.get_scores(query, docs): release_scratch() frees it)"
                if getattr(self, "_scratch_slot", None) is not None else "")
        raise RuntimeError(f"all {_lib.ERH_BM25_SLOTS} BM25 index slots of this engine are in use{hint}")

    def scratch_bm25_slot(self) -> int:
        """The engine's scratch slot for throw-away indices (BM25Retriever.get_scores(query, docs): a handful of sentences per
        call on the compressor's per-query path).  Allocated on first use and kept: re-setting an index re-uses the slot's
        device buffers, whereas alloc / free per call costs device allocations, frees and a full synchronisation each time.
        It occupies one of the ERH_BM25_SLOTS slots until release_scratch() (or close())."""
        if getattr(self, "_scratch_slot", None) is None:
            self._scratch_slot = self.alloc_bm25_slot()
        return self._scratch_slot

    def release_scratch(self):
        """Give the scratch slot (and its device buffers) back; the next get_scores(query, docs) allocates it again."""
        slot = getattr(self, "_scratch_slot", None)
        if slot is not None:
            self._scratch_slot = None
            self.free_bm25_slot(slot)

    def free_bm25_slot(self, slot: int):
        """Empty a slot: the device copies of its index are freed as well (erh_bm25_release)."""
        if self._bm25_slots[slot] is not None and self._bm25_slots[slot] is not self._RESERVED and getattr(self, "_h", None):
            self._check(self._lib.erh_bm25_release(self._h, int(slot)))
        self._bm25_slots[slot] = None
        if getattr(self, "_scratch_slot", None) == slot:
            self._scratch_slot = None

    def _select(self, slot: Optional[int]) -> int:
        slot = 0 if slot is None else int(slot)              # every BM25 call names its slot; None = slot 0, as documented
        if not 0 <= slot < _lib.ERH_BM25_SLOTS:
            raise ValueError("BM25 slot out of range")
        if slot != self._bm25_cur:
            self._check(self._lib.erh_bm25_select(self._h, slot))
            self._bm25_cur = slot
        return slot

    def set_bm25(self, index: BM25Index, payload_on_device: bool = False, slot: Optional[int] = None):
        """Upload CSR postings into `slot`.  payload_on_device=True lets the GPU evaluate IDF*TF/(TF+k1*lenNorm)."""
        slot = self._select(slot)
        indptr = _np(index.indptr, np.int64)
        doc_ids = _np(index.doc_ids, np.int32)
        if payload_on_device:
            tf = _np(index.tf, np.int32)
            dl = _np(index.doc_len, np.int32)
            idf = _np(index.idf, np.float64 if index.variant == OKAPI else np.float32)
            rc = self._lib.erh_set_bm25_tf(self._h, index.variant, index.n_vocab, index.n_docs, index.nnz,
                                           _ptr(indptr), _ptr(doc_ids), _ptr(tf), _ptr(dl), _ptr(idf),
                                           float(index.avgdl), float(index.k1), float(index.b))
        else:
            pay = _np(index.payload, np.float64 if index.variant == OKAPI else np.float32)
            if pay.shape[0] != index.nnz:
                raise ValueError("index has no host payload; build with compute_payload=True")
            rc = self._lib.erh_set_bm25_csr(self._h, index.variant, index.n_vocab, index.n_docs, index.nnz,
                                            _ptr(indptr), _ptr(doc_ids), _ptr(pay))
        self._check(rc)
        self._bm25_slots[slot] = index

    def build_bm25(self, flat, doc_lens, n_vocab: int, variant: int = OKAPI, k1: float = 1.5, b: float = 0.75,
                   epsilon: float = 0.25, slot: Optional[int] = None, fetch: bool = True) -> BM25Index:
        """Index a corpus given as one flat stream of term ids (document order) + tokens per document, ON THE DEVICE
        (erh_build_bm25_index: radix sort of (term, doc) keys, run-length tf, df, payload).  `flat` / `doc_lens` may
        be numpy arrays or device tensors (int32).  fetch=True copies the CSR arrays back into the returned BM25Index
        (tests, cpu baselines); fetch=False returns an index object that only carries the sizes and parameters."""
        slot = self._select(slot)
        if _is_torch(flat):
            import torch
            flat = flat.contiguous().to(torch.int32)
            doc_lens = doc_lens.contiguous().to(torch.int32)
            is_dev = 1 if flat.is_cuda else 0
            if is_dev and not doc_lens.is_cuda:
                doc_lens = doc_lens.to(flat.device)
            if is_dev:
                torch.cuda.current_stream(flat.device).synchronize()
            n_tok, n_docs = int(flat.shape[0]), int(doc_lens.shape[0])
        else:
            flat = _np(flat, np.int32)
            doc_lens = _np(doc_lens, np.int32)
            is_dev = 0
            n_tok, n_docs = int(flat.shape[0]), int(doc_lens.shape[0])
        nnz = C.c_int64()
        self._check(self._lib.erh_build_bm25_index(self._h, int(variant), int(n_vocab), n_docs, n_tok, _ptr(flat),
                                                   _ptr(doc_lens), is_dev, float(k1), float(b), float(epsilon),
                                                   C.byref(nnz)))
        n = int(nnz.value)
        idx = BM25Index(variant=int(variant), n_docs=n_docs, n_vocab=int(n_vocab), indptr=np.zeros(0, np.int64),
                        doc_ids=np.zeros(0, np.int32), tf=np.zeros(0, np.int32),
                        doc_len=np.zeros(0, np.int32), idf=np.zeros(0), avgdl=0.0, k1=k1, b=b, epsilon=epsilon,
                        payload=np.zeros(0), n_postings=n)
        if fetch:
            idx.indptr = np.empty(int(n_vocab) + 1, np.int64)
            idx.doc_ids = np.empty(n, np.int32)
            idx.tf = np.empty(n, np.int32)
            idf = np.empty(int(n_vocab), np.float64)
            avgdl, avg_idf = C.c_double(), C.c_double()
            self._check(self._lib.erh_get_bm25_csr(self._h, _ptr(idx.indptr), _ptr(idx.doc_ids), _ptr(idx.tf), _ptr(idf),
                                                   C.byref(avgdl), C.byref(avg_idf)))
            idx.idf = idf if variant == OKAPI else idf.astype(np.float32)
            idx.avgdl, idx.average_idf = avgdl.value, avg_idf.value
            idx.doc_len = (doc_lens.cpu().numpy() if _is_torch(doc_lens) else doc_lens).astype(np.int32)
        self._bm25_slots[slot] = idx
        if fetch:
            idx.payload = self.get_bm25_payload(slot)
        return idx

    def get_bm25_payload(self, slot: Optional[int] = None) -> np.ndarray:
        self._select(slot)
        assert self.bm25 is not None
        out = np.empty(self.bm25.nnz, np.float64 if self.bm25.variant == OKAPI else np.float32)
        self._check(self._lib.erh_get_bm25_payload(self._h, _ptr(out)))
        return out

    def set_doc_meta(self, n_docs: int, content_id=None, dir_id=None):
        cid = None if content_id is None else _np(content_id, np.int32)
        did = None if dir_id is None else _np(dir_id, np.int16)
        for name, a in (("content_id", cid), ("dir_id", did)):
            if a is not None and a.shape != (int(n_docs),):
                raise ValueError(f"{name} must have one entry per document ({n_docs}), got shape {a.shape}")
        self._check(self._lib.erh_set_doc_meta(self._h, int(n_docs), _ptr(cid), _ptr(did)))
        self.n_meta = int(n_docs)

    # -- helpers ------------------------------------------------------------------------------
    def _q_desc(self, q):
        """(array, dtype code, is_device, B); the C ABI takes no query dimension, so the shape is checked here."""
        if _is_torch(q):
            import torch
            q = q.contiguous()
            dt = {torch.float16: _lib.ERH_F16, torch.float32: _lib.ERH_F32}.get(q.dtype)
            if dt is None:
                raise TypeError("queries must be float16 or float32")
            is_dev = 1 if q.is_cuda else 0
        else:
            q = np.ascontiguousarray(q)
            if q.ndim == 1:
                q = q[None, :]
            if q.dtype == np.float16:
                dt = _lib.ERH_F16
            else:
                q = np.ascontiguousarray(q, dtype=np.float32)
                dt = _lib.ERH_F32
            is_dev = 0
        if self.d and (q.ndim != 2 or int(q.shape[1]) != self.d):   # (before set_dense the library reports the state error)
            raise ValueError(f"queries must be [B, {self.d}] (the chunk matrix's dimension), got {tuple(q.shape)}")
        return q, dt, is_dev, int(q.shape[0])

    @staticmethod
    def _filter(filter_dir, B):
        if filter_dir is None:
            return None
        f = _np(filter_dir, np.int16)
        if f.shape[0] != B:
            raise ValueError("filter_dir must have one entry per query")
        return f

    @staticmethod
    def _outs(B, k, device_out):
        if device_out:
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            return (torch.empty((B, k), dtype=torch.int32, device=dev),
                    torch.empty((B, k), dtype=torch.float64, device=dev),
                    torch.empty((B,), dtype=torch.int32, device=dev))
        return np.empty((B, k), np.int32), np.empty((B, k), np.float64), np.empty((B,), np.int32)

    @staticmethod
    def _stream(stream):
        if stream is None:
            return 0
        if hasattr(stream, "cuda_stream"):
            return int(stream.cuda_stream)
        return int(stream)

    # -- queries --------------------------------------------------------------------------------
    def dense_topk(self, q, k: int, filter_dir=None, mode: int = _lib.ERH_DENSE_EXACT, normalize_q: bool = False,
                   device_out: bool = False, stream=None):
        q, dt, is_dev, B = self._q_desc(q)
        f = self._filter(filter_dir, B)
        ids, sc, ln = self._outs(B, k, device_out)
        self._check(self._lib.erh_dense_topk(self._h, _ptr(q), dt, is_dev, 1 if normalize_q else 0, B, int(k),
                                             _ptr(f), int(mode), _ptr(ids), _ptr(sc), _ptr(ln),
                                             1 if device_out else 0, self._stream(stream)))
        return ids, sc, ln

    def bm25_topk(self, q_indptr, q_tok, k: int, filter_dir=None, device_out: bool = False, stream=None,
                  slot: Optional[int] = None):
        self._select(slot)
        q_indptr = _np(q_indptr, np.int32)
        q_tok = _np(q_tok, np.int32)
        B = q_indptr.shape[0] - 1
        f = self._filter(filter_dir, B)
        ids, sc, ln = self._outs(B, k, device_out)
        self._check(self._lib.erh_bm25_topk(self._h, _ptr(q_indptr), _ptr(q_tok), B, int(k), _ptr(f),
                                            _ptr(ids), _ptr(sc), _ptr(ln), 1 if device_out else 0,
                                            self._stream(stream)))
        return ids, sc, ln

    def bm25_scores(self, q_tok, slot: Optional[int] = None) -> np.ndarray:
        self._select(slot)
        assert self.bm25 is not None
        q_tok = _np(q_tok, np.int32)
        out = np.empty(self.bm25.n_docs, np.float64)
        self._check(self._lib.erh_bm25_scores(self._h, _ptr(q_tok), int(q_tok.shape[0]), _ptr(out)))
        return out

    def _check_ids(self, *arrays):
        """Fusion kernels index content_id[id]: ids must be -1 (padding) or a document of the uploaded metadata."""
        for a in arrays:
            if a.size and (int(a.max()) >= self.n_meta or int(a.min()) < -1):
                raise ValueError(f"fusion ids must lie in [-1, {self.n_meta}) (erh_set_doc_meta size)")

    def rrf(self, ids_a, len_a, ids_b, len_b, K: int = 60, topk: int = 256):
        ids_a = _np(ids_a, np.int32)
        ids_b = _np(ids_b, np.int32)
        self._check_ids(ids_a, ids_b)
        B = ids_a.shape[0]
        la = None if len_a is None else _np(len_a, np.int32)
        lb = None if len_b is None else _np(len_b, np.int32)
        ids, sc, ln = self._outs(B, topk, False)
        self._check(self._lib.erh_rrf(self._h, _ptr(ids_a), _ptr(la), ids_a.shape[1], _ptr(ids_b), _ptr(lb),
                                      ids_b.shape[1], B, int(K), int(topk), _ptr(ids), _ptr(sc), _ptr(ln), 0, 0))
        return ids, sc, ln

    def fusion(self, ids_a, sc_a, len_a, ids_b, sc_b, len_b, topk: int = 256):
        ids_a = _np(ids_a, np.int32)
        ids_b = _np(ids_b, np.int32)
        sc_a = _np(sc_a, np.float64)
        sc_b = _np(sc_b, np.float64)
        self._check_ids(ids_a, ids_b)
        B = ids_a.shape[0]
        la = None if len_a is None else _np(len_a, np.int32)
        lb = None if len_b is None else _np(len_b, np.int32)
        ids, sc, ln = self._outs(B, topk, False)
        self._check(self._lib.erh_fusion(self._h, _ptr(ids_a), _ptr(sc_a), _ptr(la), ids_a.shape[1],
                                         _ptr(ids_b), _ptr(sc_b), _ptr(lb), ids_b.shape[1], B, int(topk),
                                         _ptr(ids), _ptr(sc), _ptr(ln), 0, 0))
        return ids, sc, ln

    def hybrid_topk(self, q, q_indptr, q_tok, k_dense: int = 288, k_sparse: int = 192, K: int = 60, topk: int = 256,
                    filter_dir=None, normalize_q: bool = False, device_out: bool = False, stream=None,
                    slot: Optional[int] = None, filter_dense="same"):
        """filter_dir restricts the sparse route; the dense route uses filter_dense (default: the same column) --
        the reference keeps the two knobs apart (filter_dict -> BM25, filters -> Qdrant; retrievers.py:278,283)."""
        self._select(slot)
        q, dt, is_dev, B = self._q_desc(q)
        q_indptr = _np(q_indptr, np.int32)
        q_tok = _np(q_tok, np.int32)
        if q_indptr.shape[0] - 1 != B:
            raise ValueError("dense and sparse query batches differ in size")
        f = self._filter(filter_dir, B)
        fdn = f if isinstance(filter_dense, str) else self._filter(filter_dense, B)
        ids, sc, ln = self._outs(B, topk, device_out)
        self._check(self._lib.erh_hybrid_topk(self._h, _ptr(q), dt, is_dev, 1 if normalize_q else 0,
                                              _ptr(q_indptr), _ptr(q_tok), B, int(k_dense), int(k_sparse), int(K),
                                              int(topk), _ptr(f), _ptr(fdn), _ptr(ids), _ptr(sc), _ptr(ln),
                                              1 if device_out else 0, self._stream(stream)))
        return ids, sc, ln

    # -- multi-GPU exchange (one handle = one rank) ------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128-byte RCCL id; rank 0 creates it and the caller hands it to every rank (any side channel)."""
        buf = C.create_string_buffer(128)
        rc = _lib.load().erh_comm_unique_id(buf)
        if rc != 0:
            raise _lib.ErhError(rc, "erh_comm_unique_id failed (librccl.so.1 not loadable?)")
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """Join the RCCL communicator of erh_allgather_topk (collective: every rank calls it)."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._check(self._lib.erh_comm_init(self._h, int(rank), int(world), C.c_char_p(unique_id)))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        """Leave the library's RCCL communicator (also reaps one whose erh_comm_init timed out and returned later).  The
        engine's rank / world describe the JOB's sharding, not the communicator: they stay as they are."""
        self._check(self._lib.erh_comm_destroy(self._h))

    def topk_row_bytes(self, k: int) -> int:
        return int(self._lib.erh_topk_row_bytes(int(k)))

    def allgather_topk(self, ids, sc, ln, n_queries: int, out=None, stream=None):
        """Device tensors of this rank's shard -> the global [n_queries x k] result (RCCL all-gather inside the
        library, on `stream`).  `out` = preallocated (ids, scores, len) device tensors, or None to allocate."""
        import torch
        k = int(ids.shape[1])
        if out is None:
            dev = ids.device
            out = (torch.empty((n_queries, k), dtype=torch.int32, device=dev),
                   torch.empty((n_queries, k), dtype=torch.float64, device=dev),
                   torch.empty((n_queries,), dtype=torch.int32, device=dev))
        self._check(self._lib.erh_allgather_topk(self._h, _ptr(ids), _ptr(sc), _ptr(ln), int(ids.shape[0]), k,
                                                 int(n_queries), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                                 self._stream(stream)))
        return out

    def pack_topk(self, ids, sc, ln, rows_out, stream=None):
        """[b_local x k] device result -> packed rows (uint8 tensor [m, row_bytes], padding rows have len 0)."""
        self._check(self._lib.erh_pack_topk(self._h, _ptr(ids), _ptr(sc), _ptr(ln), int(ids.shape[0]),
                                            int(ids.shape[1]), int(rows_out.shape[0]), _ptr(rows_out),
                                            self._stream(stream)))

    def unpack_topk(self, gathered, n_queries: int, world: int, k: int, out, stream=None):
        self._check(self._lib.erh_unpack_topk(self._h, _ptr(gathered), int(n_queries), int(world), int(k),
                                              _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), self._stream(stream)))

    # -- measurement ------------------------------------------------------------------------------
    def sync(self, stream=None):
        self._check(self._lib.erh_sync(self._h, self._stream(stream)))

    def dense_check(self, stream=None):
        self._check(self._lib.erh_dense_check(self._h, self._stream(stream)))

    def set_profiling(self, on: bool):
        self._check(self._lib.erh_set_profiling(self._h, 1 if on else 0))

    def reset_kernel_time(self):
        self._check(self._lib.erh_reset_kernel_time(self._h))

    def kernel_time(self, cls: int):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.erh_get_kernel_time(self._h, cls, C.byref(ms), C.byref(n)))
        by, fl = C.c_double(), C.c_double()
        self._check(self._lib.erh_get_kernel_work(self._h, cls, C.byref(by), C.byref(fl)))
        return {"ms": ms.value, "launches": n.value, "bytes": by.value, "flops": fl.value}

    def set_option(self, name: str, value: int):
        self._check(self._lib.erh_set_option(self._h, name.encode(), int(value)))

    def debug_counters(self):
        out = np.zeros(16, np.uint64)
        self._check(self._lib.erh_debug_counters(self._h, _ptr(out)))
        return out

    def dense_diag(self):
        e, m, u = C.c_double(), C.c_double(), C.c_int()
        self._check(self._lib.erh_dense_diag(self._h, C.byref(e), C.byref(m), C.byref(u)))
        x = C.c_int()
        self._check(self._lib.erh_dense_exhaustive_count(self._h, C.byref(x)))
        return {"max_abs_err": e.value, "margin": m.value, "uncertified": u.value, "exhaustive": x.value}

    STAT_NAMES = ("dense_calls", "bm25_calls", "hybrid_calls", "dense_scan_pp5_launches", "dense_scan_pp3_launches",
                  "dense_scan_gemv_launches", "dense_scan_tile_launches", "dense_sample_passes", "dense_tile384_nomem",
                  "dense_exhaustive_queries", "bm25_redo_segments", "dense_block_groups", "dense_grouped_launches", "bm25_mixed_launches")

    def stat(self, name: str) -> int:
        """One counter of erh_get_stat (which kernels answered the calls since the last reset_stats)."""
        v = C.c_int64()
        self._check(self._lib.erh_get_stat(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def stats(self) -> dict:
        return {n: self.stat(n) for n in self.STAT_NAMES}

    def dense_candidates_last_call(self) -> int:
        """Candidates the last dense pipeline's scan handed to its final kernel, summed over the queries (synchronises)."""
        return self.stat("dense_candidates_last_call")

    def reset_stats(self):
        self._check(self._lib.erh_reset_stats(self._h))

    def debug_dense_scores(self, q16: np.ndarray, row0: int, rows: int, use_mfma: bool) -> np.ndarray:
        q16 = np.ascontiguousarray(q16, dtype=np.float16)
        B = q16.shape[0]
        out = np.empty((B, rows), np.float32)
        self._check(self._lib.erh_debug_dense_scores(self._h, _ptr(q16), B, int(row0), int(rows),
                                                     1 if use_mfma else 0, _ptr(out)))
        return out
