"""The speculative first pruning threshold of the dense route: erh_dense_topk seeds it with the
rank-r score of the stored prefix, r = erh_dense_seed_rank(k, n0, N) << k.  The prefix is an even sample of the corpus,
so the number of true top-k members that land in it is Binomial(k, n0 / N); the threshold is too high only if more than
r - 1 of them do (then fewer than k chunks can reach it -- which the device check catches and the exhaustive path
answers).  This pins the claimed failure probability of the rank rule on the host: no GPU, no compute calls."""
import pytest
from scipy import stats

from easyrag_amd import _lib


@pytest.fixture(scope="module")
def rank():
    return _lib.load(build_if_missing=True).erh_dense_seed_rank


@pytest.mark.parametrize("k", [1, 5, 10, 50, 100, 192, 288, 768])
@pytest.mark.parametrize("n0,n", [(32768, 1_000_000), (32768, 10_000_000), (16960, 1_000_000), (1024, 5000), (256, 60000),
                                  (2048, 20000), (32768, 40000)])
def test_rank_rule_has_a_negligible_failure_probability(rank, k, n0, n):
    r = rank(k, n0, n)
    assert 1 <= r <= k
    if r == k:
        return                                             # guaranteed bound: the k-th best of a subset, nothing to verify
    # speculation fails for a query iff at least r of the true top k are in the prefix
    p_fail = stats.binom.sf(r - 1, k, n0 / n)
    assert p_fail < 1e-7, (k, n0, n, r, p_fail)
    # and it buys something: the threshold admits r / n0 of the scores instead of k / n0
    assert r < k


def test_rank_rule_edges(rank):
    assert rank(288, 32768, 32768) == 288                  # no scan stage: nothing to speculate about
    assert rank(288, 32768, 1000) == 288
    assert rank(1, 32768, 1_000_000) == 1                  # top-1: the rule would ask for more than k
    assert rank(288, 32768, 1_000_000) == 33
    assert rank(100, 32768, 1_000_000) == 19
    assert rank(288, 0, 1_000_000) == 288
