"""Native text side of the index build (second half): erh_vocab_* against the Python dict loop it replaces,
erh_cutter_* against hand-derived segmentations and against a plain-Python restatement of jieba 0.42.1's HMM=False
algorithm (oracle/jieba_cut.py; jieba itself cannot be installed here -- unpinned, see that file).  No GPU needed."""
import json
import os
import time

import numpy as np
import pytest

from easyrag_amd.index import build_bm25_index, vocab_ids, vocab_ids_python
from easyrag_amd.text import NativeCutter, NativeVocab
from oracle.jieba_cut import DictCutter

HERE = os.path.dirname(os.path.abspath(__file__))

MINI_DICT = """\
北京 300
大学 400
北京大学 250
大学生 200
生 50
前来 80
应聘 60
研究 300
研究生 150
生命 200
命 30
起源 100
的 5000
网络 220
设备 210
网络设备 40
配置 180
T恤 10
C++ 25
"""


def test_cutter_hand_derived_cases():
    """total = 8295.  '研究生命起源': routes 研究|生命|起源 (log p = ln300+ln200+ln100 - 3 ln T) vs 研究生|命|起源
    (ln150+ln30+ln100 - 3 ln T): 300*200 > 150*30, so the first.  '北京大学生前来应聘': 北京大学|生|... scores
    250*50 = 12500 against 北京|大学生 300*200 = 60000 (two words each): the second wins."""
    c = NativeCutter(MINI_DICT)
    assert c.cut("研究生命起源") == ["研究", "生命", "起源"]
    assert c.cut("北京大学生前来应聘") == ["北京", "大学生", "前来", "应聘"]
    # unknown characters stand alone; ASCII letters / digits glue; ' ' is a token (the reference drops it later)
    # 网络设备 (40) as one word: ln 40 - ln T = -5.33 against ln 220 + ln 210 - 2 ln T = -7.30: the single word is likelier
    assert c.cut("网络设备的配置 abc123，好") == ["网络设备", "的", "配置", " ", "abc123", "，", "好"]
    assert c.cut("网络设") == ["网络", "设"]                              # 网络设 is only a prefix entry (frequency 0)
    # dictionary words with ASCII inside the han class, '.', '+', '-' are block characters but not glued
    assert c.cut("T恤3.5C++") == ["T恤", "3", ".", "5", "C++"]
    assert c.cut("a\r\nb\tc") == ["a", "\r\n", "b", "\t", "c"]
    assert c.cut("") == []


def _random_sentences(rng, words, n):
    extra = list("，。！？ abcXYZ019+#&._%-\t\n々〆ヶ") + ["\r\n", "𠀀", "é"]
    out = []
    for _ in range(n):
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            r = rng.random()
            if r < 0.65:
                parts.append(words[int(rng.integers(0, len(words)))])
            elif r < 0.85:
                parts.append(extra[int(rng.integers(0, len(extra)))])
            else:
                w = words[int(rng.integers(0, len(words)))]
                parts.append(w[: max(1, len(w) - 1)])          # a proper prefix: exercises the frequency-0 entries
        out.append("".join(parts))
    return out


def test_cutter_equals_python_restatement_on_random_text():
    rng = np.random.default_rng(3)
    chars = [chr(c) for c in range(0x4E00, 0x4E00 + 60)] + list("abAB12")
    words = sorted({"".join(chars[int(i)] for i in rng.integers(0, len(chars), size=int(rng.integers(1, 5))))
                    for _ in range(400)})
    dict_text = "\n".join(f"{w} {int(rng.integers(1, 2000))}" + (" n" if i % 3 == 0 else "") for i, w in enumerate(words))
    dict_text += "\n重复 7\n重复 9\n"                                      # a repeated entry: last frequency, both counted in total
    nat, ref = NativeCutter(dict_text), DictCutter(dict_text)
    for s in _random_sentences(rng, words + ["重复"], 3000):
        assert nat.cut(s) == ref.cut(s), repr(s)
        assert "".join(nat.cut(s)) == s                                   # tokens partition the sentence


def test_cutter_on_reference_queries_when_present():
    """The reference's 103 questions (src/data/question.jsonl) with a dictionary made from their own frequent 2- and
    3-grams: the C++ cutter and the Python restatement agree token for token."""
    path = "/root/reference/src/data/question.jsonl"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    qs = [json.loads(ln)["query"] for ln in open(path, encoding="utf-8") if ln.strip()]
    counts = {}
    for q in qs:
        for n in (2, 3):
            for i in range(len(q) - n + 1):
                g = q[i:i + n]
                if all("一" <= ch <= "鿕" for ch in g):
                    counts[g] = counts.get(g, 0) + 1
    dict_text = "\n".join(f"{g} {c * (3 if len(g) == 2 else 5)}" for g, c in sorted(counts.items()) if c >= 2)
    nat, ref = NativeCutter(dict_text), DictCutter(dict_text)
    n_multi = 0
    for q in qs:
        got = nat.cut(q)
        assert got == ref.cut(q), q
        n_multi += sum(len(t) > 1 for t in got)
    assert len(qs) == 103 and n_multi > 200


def test_cutter_lone_surrogate_is_one_character():
    """a lone surrogate in a Python str is ONE character for jieba (it walks the str); its surrogatepass encoding is
    three bytes, which the native decoder must not split into three one-byte tokens (nor overflow the output)."""
    c = NativeCutter(MINI_DICT)
    ora = DictCutter(MINI_DICT)
    for text in ("a\ud800b", "\udfff", "中\ud83d文 x\udc00"):
        assert c.cut(text) == list(ora.cut(text, HMM=False))
    assert c.cut("a\ud800b") == ["a", "\ud800", "b"]
    vocab = NativeVocab()
    flat, lens = c.encode_texts(["a\ud800b"], vocab)
    assert lens.tolist() == [3] and len(vocab) == 3


def test_cutter_rejects_bad_dictionary():
    from easyrag_amd._lib import ErhError
    for bad in ("词", "词 x1", "", "词 -3"):
        with pytest.raises(ErhError):
            NativeCutter(bad)


def test_vocab_native_equals_python_dict_loop():
    rng = np.random.default_rng(0)
    alphabet = ["的", "网络", "设备", "a", "B", "é", "𠀀", "配置", " ", "x y", "\t", "long-token-" * 8] + [f"t{i}" for i in range(300)]
    corpus = [[alphabet[int(i)] for i in rng.integers(0, len(alphabet), size=int(rng.integers(0, 30)))] for _ in range(2000)]
    v_py, flat_py, lens_py = vocab_ids_python(corpus)
    v_nat, flat, lens = vocab_ids(corpus, native=True)
    assert isinstance(v_nat, NativeVocab) and len(v_nat) == len(v_py)
    assert np.array_equal(flat, flat_py) and np.array_equal(lens, lens_py)
    for tok, i in v_py.items():
        assert v_nat[tok] == i and v_nat.token(i) == tok and tok in v_nat
    assert "never-seen" not in v_nat
    assert np.array_equal(v_nat.ids_of(["网络", "never-seen", "a", "网络"]), [v_py["网络"], v_py["a"], v_py["网络"]])
    # the index built from either id stream is the same object, bit for bit
    a, b = build_bm25_index(corpus, 1), build_bm25_index([list(d) for d in corpus], 1)
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.doc_ids, b.doc_ids) and np.array_equal(a.payload, b.payload)
    assert np.array_equal(a.tokens_to_ids(["设备", "zzz", "的"]), [v_py["设备"], v_py["的"]])


def test_vocab_falls_back_for_tokens_the_library_cannot_carry():
    corpus = [["a", "b\x00c"], ["", "a"], [1, 2, (3, 4)]]
    for doc in corpus:
        v, flat, lens = vocab_ids([doc], native=True)
        assert isinstance(v, dict) and list(lens) == [len(doc)] and len(v) == len(set(doc))


def test_vocab_native_on_a_large_corpus():
    """100k documents x 56 tokens: the native id walk equals the Python dict loop (timings: scripts/text_bench.py ->
    profiles/r03_text_native.log; for ready-made Python token lists the dict loop is the faster of the two)."""
    rng = np.random.default_rng(1)
    words = [f"w{i}" for i in range(50000)]
    ids = rng.zipf(1.3, size=5_600_000) % len(words)
    lens = np.full(100_000, 56)
    off = np.concatenate([[0], np.cumsum(lens)])
    corpus = [[words[j] for j in ids[off[i]:off[i + 1]]] for i in range(len(lens))]
    t0 = time.perf_counter()
    v_nat, flat, ln = vocab_ids(corpus, native=True)
    t_nat = time.perf_counter() - t0
    t0 = time.perf_counter()
    v_py, flat_py, ln_py = vocab_ids_python(corpus)
    t_py = time.perf_counter() - t0
    print(f"vocab ids, {flat.shape[0]} tokens / {len(corpus)} docs: native {t_nat:.2f} s, python loop {t_py:.2f} s")
    assert np.array_equal(flat, flat_py) and np.array_equal(ln, ln_py) and len(v_nat) == len(v_py)


def test_encode_texts_equals_cut_filter_and_dict_loop():
    """erh_text_encode (cut + stop words + ids in one native pass) against the same three steps in Python
    (tokenize_and_remove_stopwords of the shim, retrievers.py:72-76, then the dict loop)."""
    from easyrag_amd.retrievers import tokenize_and_remove_stopwords
    rng = np.random.default_rng(9)
    chars = [chr(c) for c in range(0x4E00, 0x4E00 + 40)]
    words = sorted({"".join(chars[int(i)] for i in rng.integers(0, len(chars), size=int(rng.integers(1, 4)))) for _ in range(200)})
    dict_text = "\n".join(f"{w} {int(rng.integers(1, 500))}" for w in words)
    cutter = NativeCutter(dict_text)
    texts = _random_sentences(rng, words, 500) + ["", " ", "的 的"]
    stop = {words[3], words[10], "，", "a", ""}
    want_tokens = [tokenize_and_remove_stopwords(cutter, t, stop) for t in texts]
    v_py, flat_py, lens_py = vocab_ids_python(want_tokens)
    vocab = NativeVocab()
    flat, lens = cutter.encode_texts(texts, vocab, stop)
    assert np.array_equal(flat, flat_py) and np.array_equal(lens, lens_py) and len(vocab) == len(v_py)
    # any number of host threads gives the same ids (chunk vocabularies merged in first-appearance order)
    for threads in (1, 2, 3, 8, 64):
        vt = NativeVocab()
        ft, lt = cutter.encode_texts(texts, vt, stop, threads=threads)
        assert np.array_equal(ft, flat_py) and np.array_equal(lt, lens_py) and len(vt) == len(v_py), threads
        assert [vt.token(i) for i in range(len(vt))] == [vocab.token(i) for i in range(len(vocab))]
    # ... also into a vocabulary that already holds terms
    half = len(texts) // 2
    vt = NativeVocab()
    f1, l1 = cutter.encode_texts(texts[:half], vt, stop, threads=4)
    f2, l2 = cutter.encode_texts(texts[half:], vt, stop, threads=5)
    assert np.array_equal(np.concatenate([f1, f2]), flat_py) and np.array_equal(np.concatenate([l1, l2]), lens_py)
    # query side: unknown tokens vanish, order and repeats stay
    q = texts[5] + "𠀀未知" + texts[5]
    ids, n = cutter.encode_texts([q], vocab, stop, add=False)
    want = [v_py[t] for t in tokenize_and_remove_stopwords(cutter, q, stop) if t in v_py]
    assert list(ids) == want and n[0] == len(want) and len(vocab) == len(v_py)


def test_cutter_survives_arbitrary_bytes():
    """Raw C-ABI callers may hand over anything: malformed UTF-8 must neither crash nor lose bytes -- the token ends are
    strictly increasing byte offsets that finish at the length of the input."""
    import ctypes as C

    from easyrag_amd import _lib
    lib = _lib.load()
    cutter = NativeCutter(MINI_DICT)
    rng = np.random.default_rng(11)
    pool = [b"\xe5\x8c\x97\xe4\xba\xac", b"\xe5\xa4\xa7\xe5\xad\xa6", b"abc", b" ", b"\r\n", b"\xff", b"\xc0\x80", b"\xe5\x8c",
            b"\xf0\x9f\x98\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"C++", b"\x00", b"\x80\x80"]
    for _ in range(300):
        raw = b"".join(pool[i] for i in rng.integers(0, len(pool), size=int(rng.integers(0, 24))))
        if rng.random() < 0.3:
            raw = bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8))
        ends = np.empty(len(raw) + 1, np.int64)
        n = C.c_int64(0)
        assert lib.erh_cutter_cut(cutter._h, raw, len(raw), ends.ctypes.data, len(ends), C.byref(n)) == 0
        e = ends[: n.value]
        assert np.all(np.diff(np.concatenate([[0], e])) > 0)
        assert (n.value == 0) == (len(raw) == 0) and (len(raw) == 0 or e[-1] == len(raw))


def _synthetic_hmm(rng, chars):
    """A random but well-formed finalseg model over `chars` (what jieba ships is trained; the algorithm does not care)."""
    start = {"B": -0.26, "E": -3.14e100, "M": -3.14e100, "S": -1.46}
    trans = {"B": {"E": -0.51, "M": -0.92}, "E": {"B": -0.59, "S": -0.81}, "M": {"E": -0.33, "M": -1.26}, "S": {"B": -0.72, "S": -0.67}}
    emit = {s: {} for s in "BMES"}
    for ch in chars:
        for s in "BMES":
            if rng.random() < 0.8:                                        # some characters are unknown to some states
                emit[s][ch] = -float(rng.integers(2, 14)) - float(rng.integers(0, 4)) / 4.0   # quarter steps: ties do happen
    return start, trans, emit


def test_cutter_hmm_mode_equals_python_restatement():
    """jieba's default call cut(sentence) = HMM=True: runs of out-of-dictionary single characters go through
    finalseg (viterbi over B M E S, then its own re_han / re_skip splitting) -- native against the restatement, on a
    dictionary that leaves most characters out so that the HMM has work to do."""
    from easyrag_amd.text import hmm_model_text
    from oracle.jieba_cut import HmmModel
    rng = np.random.default_rng(23)
    chars = [chr(c) for c in range(0x4E00, 0x4E00 + 60)]
    words = sorted({"".join(chars[int(i)] for i in rng.integers(0, 25, size=int(rng.integers(2, 4)))) for _ in range(40)})
    dict_text = "\n".join(f"{w} {int(rng.integers(1, 500))}" for w in words)
    start, trans, emit = _synthetic_hmm(rng, chars)
    native = NativeCutter(dict_text, hmm_model_text(start, trans, emit))
    ref = DictCutter(dict_text, HmmModel(start, trans, emit))
    assert native.has_hmm
    extra = list("abcXYZ0123456789+#&._%- \t\r\n，。") + ["\r\n", "3.14%", "C++", "a.b"]
    n_hmm_tokens = 0
    for _ in range(400):
        parts = []
        for _ in range(int(rng.integers(1, 12))):
            r = rng.random()
            if r < 0.3:
                parts.append(words[int(rng.integers(0, len(words)))])
            elif r < 0.8:
                parts.append("".join(chars[int(i)] for i in rng.integers(0, len(chars), size=int(rng.integers(1, 6)))))
            else:
                parts.append(extra[int(rng.integers(0, len(extra)))])
        text = "".join(parts)
        want = ref.cut(text)
        got = native.cut(text)
        assert got == want, (text, got, want)
        assert "".join(got) == text
        assert native.cut(text, HMM=False) == ref.cut(text, HMM=False)
        n_hmm_tokens += sum(1 for a, b in zip(got, native.cut(text, HMM=False)) if a != b)
    assert n_hmm_tokens > 100                                             # the HMM path really changed segmentations
    # the fused corpus pass follows the cutter's mode, on any number of threads
    texts = ["".join(chars[int(i)] for i in rng.integers(0, len(chars), size=int(rng.integers(3, 30)))) for _ in range(200)]
    want_tokens = [[w for w in ref.cut(t) if w != " "] for t in texts]
    v_py, flat_py, lens_py = vocab_ids_python(want_tokens)
    for threads in (1, 4):
        vt = NativeVocab()
        ft, lt = native.encode_texts(texts, vt, threads=threads)
        assert np.array_equal(ft, flat_py) and np.array_equal(lt, lens_py)
    # without a model HMM=True is refused, and removing the model restores the HMM=False default
    with pytest.raises(NotImplementedError):
        NativeCutter(dict_text).cut("x", HMM=True)
    native.set_hmm(None)
    assert not native.has_hmm and native.cut(texts[0]) == ref.cut(texts[0], HMM=False)
    with pytest.raises(Exception):
        native.set_hmm("emit Q 好 -1.0\n")


def test_cutter_hmm_hand_derived():
    """Two unknown characters with a model that makes 'B then E' the best path are glued, 'S S' keeps them apart."""
    from easyrag_amd.text import hmm_model_text
    d = "北京 100\n"
    start = {"B": -1.0, "S": -1.0}
    trans = {"B": {"E": -0.1, "M": -5.0}, "E": {"B": -1.0, "S": -1.0}, "M": {"E": -1.0, "M": -1.0}, "S": {"B": -1.0, "S": -1.0}}
    glue = {"B": {"甲": -1.0}, "E": {"乙": -1.0}, "M": {}, "S": {"甲": -9.0, "乙": -9.0}}
    apart = {"B": {"甲": -9.0}, "E": {"乙": -9.0}, "M": {}, "S": {"甲": -1.0, "乙": -1.0}}
    assert NativeCutter(d, hmm_model_text(start, trans, glue)).cut("北京甲乙北京") == ["北京", "甲乙", "北京"]
    assert NativeCutter(d, hmm_model_text(start, trans, apart)).cut("北京甲乙北京") == ["北京", "甲", "乙", "北京"]
    assert NativeCutter(d).cut("北京甲乙北京") == ["北京", "甲", "乙", "北京"]
