"""Seeded synthetic workloads of the BASELINE.json shapes.

No corpus ships with the reference (it is downloaded by scripts/process.sh) and there is no network, so
benches and parity tests run on synthetic data of the documented shape:
  dense    unit-norm rows: standard normal -> L2-normalise -> fp16; queries = noisy copies of random rows
  sparse   token-id documents: Zipf(1.07) over V ids with the 64 most frequent ids removed (stop-words),
           document length ~ round(lognormal(ln 56, 0.4)) clipped to [8, 256]; a query = 8 tokens sampled
           from a target document plus 2 random ids
Small cases are generated with numpy; the 1M-chunk cases with torch on the GPU (plumbing only).

A second pair of shapes answers "is the timing an artefact of isotropic data?" (VERDICT r5, 6):
  clustered dense   topic-sorted rows x = sqrt(.4) m + sqrt(.3) c_topic + sqrt(.3) noise (unit m, unit topic centres): two chunks of
                    different topics have cosine ~0.4, of one topic ~0.7 -- what instruction-tuned embedding models produce --, and
                    2 % of the rows are exact copies of another row of their topic; queries are noisy copies of corpus rows
  reference-length  token queries whose lengths follow the reference's 103 real questions (REF_QUESTION_LENGTHS: 4 ... 45 tokens,
                    mean 10.8) instead of a fixed 10
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def dense_corpus(n: int, d: int, seed: int = 1, dtype=np.float16) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(dtype)


def dense_queries(x: np.ndarray, b: int, seed: int = 2, noise: float = 0.75) -> np.ndarray:
    """Noisy copies of random corpus rows (cos ~ 0.8 at noise 0.75), unit norm, float32."""
    rng = np.random.default_rng(seed)
    n, d = x.shape
    rows = rng.integers(0, n, size=b)
    q = x[rows].astype(np.float32) + noise * rng.standard_normal((b, d), dtype=np.float32) / np.sqrt(d)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def zipf_cdf(v: int, s: float = 1.07, drop_top: int = 64) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, v + drop_top + 1, dtype=np.float64), s)
    w = w[drop_top:]
    c = np.cumsum(w)
    return c / c[-1]


def token_corpus(n: int, v: int, seed: int = 3, mean_len: float = 56.0, sigma: float = 0.4) -> Tuple[np.ndarray, np.ndarray]:
    """Returns (flat token ids int64[T], doc_lens int64[n])."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.lognormal(np.log(mean_len), sigma, size=n)), 8, 256).astype(np.int64)
    cdf = zipf_cdf(v)
    flat = np.searchsorted(cdf, rng.random(int(lens.sum())), side="left").astype(np.int64)
    np.minimum(flat, v - 1, out=flat)
    return flat, lens


# token counts of the reference's 103 questions (src/data/question.jsonl as extracted by tests/golden/make_ref_fixture.py:
# `len(q["tokens"])` of tests/golden/ref_queries.json, sorted) -- the length distribution of the real workload's queries
REF_QUESTION_LENGTHS = (4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8,
                        9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11, 11,
                        11, 12, 12, 12, 12, 12, 12, 12, 12, 12, 13, 13, 13, 13, 13, 14, 14, 15, 16, 16, 17, 18, 18, 20, 20, 22, 22, 28, 30,
                        31, 45)


def token_queries(flat: np.ndarray, lens: np.ndarray, v: int, b: int, seed: int = 4, from_doc: int = 8,
                  random_extra: int = 2, lengths=None) -> List[np.ndarray]:
    """`lengths`: a sequence of query lengths to draw from (e.g. REF_QUESTION_LENGTHS): a query of L tokens takes round(0.8 L) from
    its target document (with repeats when the document is shorter, as real questions repeat words) and the rest at random."""
    rng = np.random.default_rng(seed)
    off = np.zeros(lens.shape[0] + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    out = []
    for _ in range(b):
        dsel = int(rng.integers(0, lens.shape[0]))
        toks = flat[off[dsel]:off[dsel + 1]]
        if lengths is None:
            pick = rng.choice(toks, size=min(from_doc, toks.shape[0]), replace=False)
            extra = rng.integers(0, v, size=random_extra)
        else:
            total = int(lengths[int(rng.integers(0, len(lengths)))])
            n_doc = max(1, int(round(0.8 * total)))
            pick = rng.choice(toks, size=n_doc, replace=n_doc > toks.shape[0])
            extra = rng.integers(0, v, size=max(0, total - n_doc))
        out.append(np.concatenate([pick, extra]).astype(np.int32))
    return out


def split_docs(flat: np.ndarray, lens: np.ndarray) -> List[np.ndarray]:
    off = np.zeros(lens.shape[0] + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return [flat[off[i]:off[i + 1]] for i in range(lens.shape[0])]


# ---- GPU-side generators for the 1M-chunk configurations (torch is plumbing here) --------------------------
def dense_corpus_torch(n: int, d: int, seed: int, device) -> "object":
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float16, device=device)
    step = 65536
    for s in range(0, n, step):
        m = min(step, n - s)
        x = torch.randn((m, d), generator=g, device=device, dtype=torch.float32)
        x = x / x.norm(dim=1, keepdim=True)
        out[s:s + m] = x.to(torch.float16)
    return out


def clustered_corpus_torch(n: int, d: int, seed: int, device, topics: int = 2000, mean2: float = 0.4, topic2: float = 0.3,
                           dup_frac: float = 0.02):
    """Anisotropic, topic-sorted unit rows (see the module docstring): chunk i belongs to topic i * topics // n."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    m = torch.randn((1, d), generator=g, device=device, dtype=torch.float32)
    m = m / m.norm()
    c = torch.randn((topics, d), generator=g, device=device, dtype=torch.float32)
    c = c / c.norm(dim=1, keepdim=True)
    a, b_, s = mean2 ** 0.5, topic2 ** 0.5, max(0.0, 1.0 - mean2 - topic2) ** 0.5
    out = torch.empty((n, d), dtype=torch.float16, device=device)
    step = 65536
    for r0 in range(0, n, step):
        rows = min(step, n - r0)
        t = (torch.arange(r0, r0 + rows, device=device, dtype=torch.int64) * topics) // n
        noise = torch.randn((rows, d), generator=g, device=device, dtype=torch.float32)
        noise = noise / noise.norm(dim=1, keepdim=True)
        x = a * m + b_ * c[t] + s * noise
        x = x / x.norm(dim=1, keepdim=True)
        out[r0:r0 + rows] = x.to(torch.float16)
    n_dup = int(n * dup_frac)
    if n_dup > 0:                                       # exact copies of another row of the same topic (a few hundred rows away)
        dst = torch.randperm(n, generator=g, device=device)[:n_dup]
        per = max(2, n // topics)
        src = (dst // per) * per + torch.randint(0, per, (n_dup,), generator=g, device=device)
        src = torch.clamp(src, max=n - 1)
        out[dst] = out[src]
    return out


def dense_queries_torch(x, b: int, seed: int, noise: float = 0.75):
    import torch
    g = torch.Generator(device=x.device)
    g.manual_seed(seed)
    n, d = x.shape
    rows = torch.randint(0, n, (b,), generator=g, device=x.device)
    q = x[rows].float() + noise * torch.randn((b, d), generator=g, device=x.device) / (d ** 0.5)
    q = q / q.norm(dim=1, keepdim=True)
    return q.to(torch.float16)


def token_csr_torch(n: int, v: int, seed: int, device, mean_len: float = 56.0, sigma: float = 0.4):
    """Zipfian token corpus -> CSR postings, built with torch sorts on the GPU.
    Returns numpy (indptr int64[V+1], doc_ids int32[nnz], tf int32[nnz], doc_lens int64[n]) plus the flat
    token stream offsets needed to draw queries (flat int64 numpy)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.clamp(torch.round(torch.exp(torch.randn(n, generator=g, device=device) * sigma + float(np.log(mean_len)))),
                       8, 256).to(torch.int64)
    total = int(lens.sum().item())
    cdf = torch.from_numpy(zipf_cdf(v)).to(device)
    flat = torch.searchsorted(cdf, torch.rand(total, generator=g, device=device, dtype=torch.float64))
    flat = torch.clamp(flat, max=v - 1)
    doc_of = torch.repeat_interleave(torch.arange(n, device=device), lens)
    key = flat * n + doc_of
    ukey, tf = torch.unique(key, return_counts=True)          # sorted: (term asc, doc asc)
    term = ukey // n
    doc = (ukey - term * n).to(torch.int32)
    df = torch.bincount(term, minlength=v)
    indptr = torch.zeros(v + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(df, 0)
    return (indptr.cpu().numpy(), doc.cpu().numpy(), tf.to(torch.int32).cpu().numpy(), lens.cpu().numpy(),
            flat.cpu().numpy())
