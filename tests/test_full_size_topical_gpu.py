"""100 M TOPICAL rows (synth_dist 4: Zipf-sized clusters of cosine 0.5 .. 0.95 — what the ladder behind the certificates is for) on
one GPU: the paths that only exist at this size — the bounded exact pass of single queries on the packed 5-bit shadow (from 40 Mi
rows), seeded by a packed-stream search over 1/32 of the rows; demotion at 18 % failures; batches whose thresholds the batch
feedback has deepened; the optional FP6 first filter and its self-suspension — must return what the EXACT PASS over all f32 rows
returns (force_fallback = 1: scan_exact_kernel, a different kernel over different bytes, itself held against the oracle at this size
by test_full_size_gpu.py), bit for bit; three queries also against the C oracle's own scan of the 100 M generated rows.
Needs ~250 GB of free HBM like test_full_size_gpu.py (fails, not skips, on a smaller card unless DAWN_ALLOW_SMALL_GPU=1)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402

N = 100_000_000
QROW0 = 1 << 40


@pytest.fixture(scope="module")
def topical(dawn):
    import torch
    free, total = torch.cuda.mem_get_info(0)
    if free < 250e9:
        msg = f"needs ~250 GB of free HBM (100 M f32 rows + int8, packed and FP6 shadows); this card has {free / 1e9:.0f} GB free"
        if os.environ.get("DAWN_ALLOW_SMALL_GPU") == "1":
            pytest.skip(msg)
        pytest.fail(msg)
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", 4)
    idx.fill_synthetic(1, 0, N, 1)
    yield idx
    idx.close()


def _queries(nq):
    return np.concatenate([synth.unit_rows_topical(1, QROW0 + 256 * i, 1) for i in range(nq)])


def test_100m_topical_single_queries_through_the_ladder_equal_the_exact_pass(dawn, oracle, topical):
    idx = topical
    Q = _queries(72)
    idx.set_option("force_fallback", 1)  # the exact pass over all f32 rows answers every query
    want = {k: [idx.search(q, k) for q in Q[:48 if k == 10 else 24]] for k in (10, 20)}
    idx.set_option("force_fallback", 0)
    s0 = idx.stats()
    for k in (10, 20):
        for b, (wl, wd) in enumerate(want[k]):
            lab, dist = idx.search(Q[b], k)
            assert np.array_equal(lab, wl) and np.array_equal(dist.view(np.uint32), wd.view(np.uint32)), (k, b)
    s1 = idx.stats()
    # the ladder did the work: certificates failed, the index was demoted on the way, no exact pass
    assert s1["fallbacks"] == s0["fallbacks"] and s1["bounded"] - s0["bounded"] >= 20 and s1["demoted"] - s0["demoted"] >= 8, (s0, s1)
    # the same with the bounded pass on the int8 shadow / unseeded / without feedback
    for name, v in (("bounded_packed", 0), ("bounded_seed", 0), ("ladder_feedback", 0), ("ladder_feedback", 2)):
        idx.set_option(name, v)
        for b in (0, 5, 11, 30, 47):
            lab, dist = idx.search(Q[b], 10)
            assert np.array_equal(lab, want[10][b][0]) and np.array_equal(dist.view(np.uint32), want[10][b][1].view(np.uint32)), (name, v, b)
    idx.set_option("bounded_packed", 1)
    idx.set_option("bounded_seed", 1)
    idx.set_option("ladder_feedback", 1)
    assert idx.stats()["fallbacks"] == s0["fallbacks"]
    # three queries against the C oracle's own scan of all 100 M generated rows
    ol, od = oracle.scan_topk_synth(1, 0, N, 1, Q[:3], 20, dist=4)
    for b in range(3):
        assert np.array_equal(want[20][b][0], ol[b]) and np.array_equal(want[20][b][1].view(np.uint32), od[b].view(np.uint32)), b


def test_100m_topical_batches_through_the_ladder_equal_the_exact_pass(dawn, topical):
    idx = topical
    Q = _queries(256)
    k = 10
    idx.set_option("f6_shadow", 0)  # (the default is auto: first the int8 pass and its own feedback alone)
    idx.set_option("force_fallback", 1)
    wl, wd, _ = idx.search_batch(Q[:64], k)  # (64 exact passes of 33 ms)
    idx.set_option("force_fallback", 0)
    s0, f0 = idx.stats(), idx.stats_batch_feedback()
    for it in range(6):  # (the batch feedback deepens the thresholds after four batches)
        lab, dist, found = idx.search_batch(Q, k)
        assert np.all(found == k)
        assert np.array_equal(lab[:64], wl) and np.array_equal(dist[:64].view(np.uint32), wd.view(np.uint32)), it
        if it == 0:
            first = (lab.copy(), dist.copy())
        assert np.array_equal(lab, first[0]) and np.array_equal(dist.view(np.uint32), first[1].view(np.uint32)), it
    s1, f1 = idx.stats(), idx.stats_batch_feedback()
    assert s1["fallbacks"] == s0["fallbacks"] and s1["bounded"] - s0["bounded"] >= 6 * 40, (s0, s1)
    assert f1["deepened_batches"] - f0["deepened_batches"] == 2, (f0, f1)
    # the second pass with exact-derived thresholds, and the FP6 first filter (suspends itself on these rows), answer alike
    idx.set_option("batch_rerun", 2)
    lab, dist, _ = idx.search_batch(Q, k)
    assert np.array_equal(lab, first[0]) and np.array_equal(dist.view(np.uint32), first[1].view(np.uint32))
    assert idx.stats_batch_feedback()["rerun_answers"] > 0
    idx.set_option("batch_rerun", 0)
    idx.set_option("f6_shadow", 1)
    try:
        for it in range(6):
            lab, dist, _ = idx.search_batch(Q, k)
            assert np.array_equal(lab, first[0]) and np.array_equal(dist.view(np.uint32), first[1].view(np.uint32)), it
        f2 = idx.stats_batch_feedback()
        assert 1 <= f2["f6_batches"] <= 4 and f2["f6_suspended"] >= 2, f2  # (more than half of its first batch in the ladder: suspended at once)
    finally:
        idx.set_option("f6_shadow", 2)
    assert idx.stats()["fallbacks"] == s0["fallbacks"]
