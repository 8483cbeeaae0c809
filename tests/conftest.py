import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def _have_gpu() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs the whole suite on a CPU box they are skipped loudly
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run through gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def engine():
    """One RetrievalEngine (libeasyrag_hip handle) for the GPU test session."""
    from easyrag_amd.engine import RetrievalEngine
    eng = RetrievalEngine(0)
    yield eng
    eng.close()


class WhitespaceTokenizer:
    """Stand-in for jieba.Tokenizer in tests: `.cut(text)` yields whitespace-separated tokens and the
    spaces between them (jieba also emits ' ' tokens, which the reference filters, retrievers.py:75)."""

    def cut(self, text):
        out = []
        for i, w in enumerate(text.split(" ")):
            if i:
                out.append(" ")
            if w:
                out.append(w)
        return out


@pytest.fixture(scope="session")
def tokenizer():
    return WhitespaceTokenizer()


def make_text_corpus(n_docs: int, vocab: int, seed: int, min_len=4, max_len=40):
    rng = np.random.default_rng(seed)
    words = [f"w{i}" for i in range(vocab)]
    p = 1.0 / np.arange(1, vocab + 1) ** 1.07
    p /= p.sum()
    docs = []
    for _ in range(n_docs):
        ln = int(rng.integers(min_len, max_len + 1))
        docs.append(" ".join(rng.choice(words, size=ln, p=p)))
    return docs
