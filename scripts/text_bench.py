#!/usr/bin/env python
"""Host timing of the native text side of the index build (csrc/text.hip) beside the Python loops it replaces:
token lists -> ids (erh_vocab_encode vs the dict loop), and text -> cut -> stop words -> ids (erh_text_encode vs the
same three steps in Python with the restated cutter).  No GPU.  Usage: python scripts/text_bench.py [n_docs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easyrag_amd.index import vocab_ids, vocab_ids_python  # noqa: E402
from easyrag_amd.text import NativeCutter, NativeVocab  # noqa: E402


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    rng = np.random.default_rng(1)
    chars = [chr(c) for c in range(0x4E00, 0x4E00 + 3000)]
    words = sorted({"".join(chars[int(i)] for i in rng.integers(0, len(chars), size=int(rng.integers(1, 4)))) for _ in range(60000)})
    ids = rng.zipf(1.2, size=n_docs * 56) % len(words)
    corpus = [[words[j] for j in ids[i * 56:(i + 1) * 56]] for i in range(n_docs)]
    t0 = time.perf_counter(); v, flat, lens = vocab_ids(corpus); t_nat = time.perf_counter() - t0
    t0 = time.perf_counter(); vp, fp, lp = vocab_ids_python(corpus); t_py = time.perf_counter() - t0
    assert np.array_equal(flat, fp) and np.array_equal(lens, lp)
    print(f"token lists -> ids   {n_docs} docs / {flat.shape[0]} tokens / {len(vp)} terms: native {t_nat:.2f} s "
          f"(of which the per-document join + encode in Python dominate), python dict loop {t_py:.2f} s")
    dict_text = "\n".join(f"{w} {int(rng.integers(1, 5000))}" for w in words)
    cutter = NativeCutter(dict_text)
    texts = ["".join(d) for d in corpus]
    stop = set(words[:50])
    t0 = time.perf_counter()
    vocab = NativeVocab()
    f2, l2 = cutter.encode_texts(texts, vocab, stop, threads=1)
    t_nat = time.perf_counter() - t0
    for threads in (2, 4, 8, 16, 32, 64):
        if threads > 2 * (os.cpu_count() or 1):
            break
        t0 = time.perf_counter()
        vt = NativeVocab()
        ft, lt = cutter.encode_texts(texts, vt, stop, threads=threads)
        dt = time.perf_counter() - t0
        assert np.array_equal(ft, f2) and np.array_equal(lt, l2) and len(vt) == len(vocab)
        print(f"erh_text_encode_mt, {threads:2d} threads: {dt:.2f} s (same ids; of which the Python side -- utf-8 encode and join of "
              f"{n_docs} strings -- is the same in every row)")
    sample = texts[: max(1, n_docs // 20)]
    t0 = time.perf_counter()
    toks = [[w for w in cutter.cut(t) if w not in stop and w != " "] for t in sample]   # (the native cut, Python around it)
    _ = vocab_ids_python(toks)
    t_py = (time.perf_counter() - t0) * (n_docs / len(sample))
    print(f"texts -> cut -> stop words -> ids   {n_docs} texts / {f2.shape[0]} tokens kept / {len(vocab)} terms: "
          f"erh_text_encode (one thread) {t_nat:.2f} s; per-text cut() + Python filter + dict loop {t_py:.2f} s "
          f"(measured on {len(sample)} texts, scaled)")


if __name__ == "__main__":
    main()
