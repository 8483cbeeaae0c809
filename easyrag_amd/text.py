"""Native text side of the BM25 index build (libeasyrag_hip.so: csrc/text.hip; include/easyrag_hip.h, erh_vocab_* /
erh_cutter_*).  Host code, no GPU involved.

  NativeVocab    token -> term id, ids by first appearance: replaces the per-token Python dict loop of the shim's index
                 build (``easyrag_amd.index.vocab_ids``) -- the reference's libraries do the same walk inside
                 ``rank_bm25.BM25Okapi.__init__`` / ``bm25s.BM25.index`` (retrievers.py:94-118).
  NativeCutter   a tokenizer object with jieba's ``cut`` interface (what ``tokenize_and_remove_stopwords`` calls,
                 retrievers.py:72-76; the pipeline passes ``jieba.Tokenizer()``, pipeline.py:176-178) running jieba
                 0.42.1's dictionary-DAG algorithm for ``cut(sentence, cut_all=False, HMM=False)`` over a caller-supplied
                 dictionary in jieba's text format, and jieba's default call (HMM=True: runs of out-of-dictionary single
                 characters re-cut by finalseg's viterbi) once the caller has supplied the model tables that ship with jieba
                 (``NativeCutter(dict_text, hmm_text)`` / ``set_hmm``; INTEGRATION.md shows the dump).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Hashable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

_SEP = "\x00"


class NativeVocab:
    def __init__(self):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.erh_vocab_create(C.byref(h))
        if rc != 0:
            raise _lib.ErhError(rc, "erh_vocab_create failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.erh_vocab_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.erh_vocab_size(self._h))

    @staticmethod
    def representable(doc: Sequence[Hashable]) -> bool:
        """Token lists go through the library as NUL-separated UTF-8; that covers lists of str without NUL / empty tokens."""
        return all(isinstance(t, str) for t in doc) and "" not in doc and _SEP not in "".join(doc)

    def encode(self, corpus: Sequence[Sequence[str]], add: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """Token lists -> (flat int32 ids of all documents back to back, int32 tokens per document).  One C call; the
        Python work is one join + one encode per DOCUMENT."""
        blobs = [_SEP.join(d).encode("utf-8", "surrogatepass") for d in corpus]
        off = np.zeros(len(blobs) + 1, np.int64)
        if blobs:
            np.cumsum(np.fromiter((len(b) for b in blobs), dtype=np.int64, count=len(blobs)), out=off[1:])
        blob = b"".join(blobs)
        lens = np.zeros(max(len(blobs), 1), np.int32)
        cap = int(off[-1]) // 2 + len(blobs) + 1                         # >= the token count unless tokens are single bytes
        need = C.c_int64(0)
        for _ in range(2):
            ids = np.empty(max(cap, 1), np.int32)
            rc = self._lib.erh_vocab_encode(self._h, blob, off.ctypes.data, len(blobs), 0, 1 if add else 0,
                                            ids.ctypes.data, cap, lens.ctypes.data, C.byref(need))
            if rc == _lib.ERH_ERR_OVERFLOW:
                cap = int(need.value)
                continue
            if rc != 0:
                raise _lib.ErhError(rc, "erh_vocab_encode failed")
            return ids[: int(need.value)].copy(), lens[: len(blobs)].copy()
        raise RuntimeError("erh_vocab_encode: token count changed between two passes")

    def ids_of(self, tokens: Sequence[str]) -> np.ndarray:
        """Query side: known tokens -> ids in order, repeats kept, out-of-vocabulary tokens dropped."""
        toks = [t for t in tokens if isinstance(t, str) and t != "" and _SEP not in t]
        if not toks:
            return np.zeros(0, np.int32)
        ids, _ = self.encode([toks], add=False)
        return ids[ids >= 0]

    def token(self, i: int) -> str:
        p, n = C.c_void_p(), C.c_int32()
        rc = self._lib.erh_vocab_token(self._h, int(i), C.byref(p), C.byref(n))
        if rc != 0:
            raise IndexError(i)
        return C.string_at(p, n.value).decode("utf-8", "surrogatepass")

    def __contains__(self, tok) -> bool:
        return isinstance(tok, str) and self.ids_of([tok]).size == 1

    def __getitem__(self, tok) -> int:
        ids = self.ids_of([tok]) if isinstance(tok, str) else np.zeros(0, np.int32)
        if ids.size != 1:
            raise KeyError(tok)
        return int(ids[0])


class NativeCutter:
    """``cut(text)`` like ``jieba.Tokenizer().cut(text)`` over the given dictionary (jieba text format: one ``word freq
    [tag]`` per line): HMM=False without a model, jieba's default HMM=True with one (``hmm_text`` / ``set_hmm``)."""

    def __init__(self, dict_text: str, hmm_text: Optional[str] = None):
        self._lib = _lib.load()
        raw = dict_text.encode("utf-8")
        h = C.c_void_p()
        rc = self._lib.erh_cutter_create(raw, len(raw), C.byref(h))
        if rc != 0:
            raise _lib.ErhError(rc, "erh_cutter_create failed (dictionary lines must read 'word freq [tag]')")
        self._h = h
        if hmm_text is not None:
            self.set_hmm(hmm_text)

    @classmethod
    def from_file(cls, path: str, hmm_path: Optional[str] = None) -> "NativeCutter":
        hmm = None
        if hmm_path is not None:
            with open(hmm_path, encoding="utf-8") as f:
                hmm = f.read()
        with open(path, encoding="utf-8") as f:
            return cls(f.read(), hmm)

    def set_hmm(self, hmm_text: Optional[str]):
        """jieba.finalseg's model as text lines ``start S lp`` / ``trans S S lp`` / ``emit S char lp`` (see
        `hmm_model_text`); with it ``cut(text)`` behaves like jieba's default ``cut(text, HMM=True)``.  None removes it."""
        raw = (hmm_text or "").encode("utf-8", "surrogatepass")
        rc = self._lib.erh_cutter_set_hmm(self._h, raw, len(raw))
        if rc != 0:
            raise _lib.ErhError(rc, "erh_cutter_set_hmm failed (lines: 'start S lp', 'trans S S lp', 'emit S char lp')")

    @property
    def has_hmm(self) -> bool:
        return bool(self._lib.erh_cutter_has_hmm(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.erh_cutter_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cut(self, text: str, cut_all: bool = False, HMM: Optional[bool] = None) -> List[str]:
        """jieba.Tokenizer.cut(sentence, cut_all=False, HMM=...).  HMM=None: True when a model is set (jieba's default call),
        else False."""
        if cut_all:
            raise NotImplementedError("NativeCutter implements jieba's cut(sentence, cut_all=False)")
        if HMM and not self.has_hmm:
            raise NotImplementedError("HMM=True needs jieba's finalseg model: NativeCutter(dict_text, hmm_text) / set_hmm()")
        raw = text.encode("utf-8", "surrogatepass")
        cap = len(raw) + 1                                   # never more tokens than bytes
        ends = np.empty(cap, np.int64)
        n = C.c_int64(0)
        mode = -1 if HMM is None else (1 if HMM else 0)
        rc = self._lib.erh_cutter_cut_mode(self._h, raw, len(raw), mode, ends.ctypes.data, cap, C.byref(n))
        if rc != 0:
            raise _lib.ErhError(rc, "erh_cutter_cut_mode failed")
        out, b = [], 0
        for e in ends[: n.value].tolist():
            out.append(raw[b:e].decode("utf-8", "surrogatepass"))
            b = e
        return out

    lcut = cut

    def encode_texts(self, texts: Sequence[str], vocab: NativeVocab, stopwords: Iterable[str] = (), add: bool = True,
                     threads: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        """``[tokenize_and_remove_stopwords(self, t, stopwords) for t in texts]`` -> term ids, without a Python object per
        token (erh_text_encode): returns (flat int32 ids, int32 tokens per text)."""
        stop = NativeVocab()
        sw = [w for w in stopwords if isinstance(w, str) and w != "" and _SEP not in w]
        if sw:
            stop.encode([sw], add=True)
        blobs = [t.encode("utf-8", "surrogatepass") for t in texts]
        off = np.zeros(len(blobs) + 1, np.int64)
        if blobs:
            np.cumsum(np.fromiter((len(b) for b in blobs), dtype=np.int64, count=len(blobs)), out=off[1:])
        blob = b"".join(blobs)
        lens = np.zeros(max(len(blobs), 1), np.int32)
        cap = int(off[-1]) + 1
        need = C.c_int64(0)
        ids = np.empty(cap, np.int32)
        if threads is None:                      # corpus side: the host's cores (the ids do not depend on the thread count)
            threads = min(os.cpu_count() or 1, 64) if add else 1
        rc = self._lib.erh_text_encode_mt(self._h, vocab._h, stop._h if sw else None, blob, off.ctypes.data, len(blobs),
                                          1 if add else 0, max(int(threads), 1), ids.ctypes.data, cap, lens.ctypes.data,
                                          C.byref(need))
        if rc != 0:
            raise _lib.ErhError(rc, "erh_text_encode_mt failed")
        return ids[: int(need.value)].copy(), lens[: len(blobs)].copy()


def hmm_model_text(start_p, trans_p, emit_p) -> str:
    """jieba.finalseg's three tables -> the text NativeCutter.set_hmm takes.  With jieba installed:

        from jieba.finalseg import prob_start, prob_trans, prob_emit
        text = hmm_model_text(prob_start.P, prob_trans.P, prob_emit.P)
    """
    lines = [f"start {s} {float(p)!r}" for s, p in start_p.items()]
    lines += [f"trans {a} {b} {float(p)!r}" for a, row in trans_p.items() for b, p in row.items()]
    lines += [f"emit {s} {ch} {float(p)!r}" for s, row in emit_p.items() for ch, p in row.items()]
    return "\n".join(lines) + "\n"
