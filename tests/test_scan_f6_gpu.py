"""GPU parity tests of the FP6 (e2m3) first filter of batched searches (dawnsearch_amd/csrc/scan_f6.hip; option "f6_shadow").

A batch on a large index can filter on a 6-bit floating-point copy of the rows first — v_mfma_scale_f32_16x16x128_f8f6f4 runs at
1.5 x the int8 matrix rate under the chip's power envelope — and re-score the survivors on the int8 shadow; the common tail and its
certificates take over from there.  Held here, forced on small indexes (option "f6_min_rows" = 0): the FP6 scores are UPPER BOUNDS
of the exact scores on every kind of row, and the search results equal the CPU oracle's (oracle/dawn_oracle.c:
src/search/vector.rs:128-134 + exact top-k) bit for bit, like every other path.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from dawnsearch_amd import synth  # noqa: E402

from test_scan_gpu import _adversarial_rows, _assert_same  # noqa: E402


def _mk(dawn, n, dtype="f32", dist=0):
    idx = dawn.VectorIndex(0, dtype=dtype)
    idx.set_option("f6_min_rows", 0)
    idx.set_option("f6_shadow", 1)
    if dist:
        idx.set_option("synth_dist", dist)
    idx.fill_synthetic(1, 0, n, 1)
    return idx


@pytest.mark.parametrize("n", [1, 15, 16, 17, 33, 1000, 8192])
def test_f6_scores_are_upper_bounds(dawn, n):
    """ub = s s_q acc + E + K2 >= x.q for every row and query (exact integer-like accumulation, measured E and ||dq||), and not
    absurdly loose (< 0.16 on unit vectors)."""
    idx = _mk(dawn, n)
    x = synth.unit_rows(1, 0, n)
    Q = np.concatenate([synth.unit_rows(2, 0, 20), synth.planted_queries(1, [n // 2], 4)])
    onehot = np.zeros((1, 384), np.float32)
    onehot[0, 5] = 1.0
    Q = np.concatenate([Q, onehot, -x[:1]])
    ub = idx.debug_f6_scores(Q)
    assert ub.shape == (len(Q), n)
    exact = Q.astype(np.float64) @ x.astype(np.float64).T
    slack = ub.astype(np.float64) - exact
    assert slack.min() > -4e-6, slack.min()
    assert slack.max() < 0.16, slack.max()


def test_f6_bounds_on_adversarial_and_topical_rows(dawn):
    rows, base, extra = _adversarial_rows(3000)
    rows = rows[:8192]
    idx = dawn.VectorIndex(0)
    idx.set_option("f6_min_rows", 0)
    idx.set_option("f6_shadow", 1)
    idx.add_batch(np.arange(1, len(rows) + 1, dtype=np.uint64), rows)
    Q = np.stack([synth.unit_rows(2, 0, 1)[0], extra[7], base[9], -base[9]])
    ub = idx.debug_f6_scores(Q)
    exact = Q.astype(np.float64) @ rows.astype(np.float64).T
    assert (ub.astype(np.float64) - exact).min() > -4e-6
    idx2 = _mk(dawn, 8192, dist=4)
    x2 = synth.unit_rows_topical(1, 0, 8192)
    Q2 = synth.unit_rows_topical(1, 1 << 40, 8)
    ub2 = idx2.debug_f6_scores(Q2)
    assert (ub2.astype(np.float64) - Q2.astype(np.float64) @ x2.astype(np.float64).T).min() > -4e-6


@pytest.mark.parametrize("n", [8193, 20_000, 300_001])
@pytest.mark.parametrize("k", [10, 20, 64])
def test_f6_batched_search_matches_oracle(dawn, oracle, n, k):
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 30), synth.planted_queries(1, [0, n // 2, n - 1], 9)])
    lab, dist, found = idx.search_batch(Q, k)
    for b, q in enumerate(Q):
        assert found[b] == k
        _assert_same(lab[b], dist[b], *oracle.scan_topk(x, ids, q, k, threads=8))
    # identical to the int8 pass's answers, and the shadow's memory goes back when it is switched off
    m1 = idx.memory()["shadows"]
    idx.set_option("f6_shadow", 0)
    assert m1 - idx.memory()["shadows"] >= n * 288
    lab0, dist0, _ = idx.search_batch(Q, k)
    assert np.array_equal(lab, lab0) and np.array_equal(dist.view(np.uint32), dist0.view(np.uint32))
    assert idx.stats()["fallbacks"] == 0


def test_f6_batch_of_256_on_1m_rows_and_adds(dawn, oracle):
    n = 1_000_000
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(3, 0, 256)
    Q[7] = synth.planted_queries(1, [4242], 5)[0]
    lab, dist, found = idx.search_batch(Q, 10)
    for b in list(range(0, 256, 17)) + [7, 255]:
        _assert_same(lab[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 10, threads=8))
    assert lab[7][0] == 4243
    st = idx.stats()
    assert st["fallbacks"] == 0 and st["bounded"] <= 2, st
    # rows added afterwards are in the FP6 shadow too (the last partial tile is re-quantised)
    extra = synth.unit_rows(9, 0, 37)
    idx.add_batch(np.arange(n + 1, n + 38, dtype=np.uint64), extra)
    Q2 = np.stack([extra[3], extra[36], Q[0]])
    lab2, dist2, _ = idx.search_batch(Q2, 10)
    xx = np.concatenate([x, extra])
    ii = np.arange(1, n + 38, dtype=np.uint64)
    for b in range(3):
        _assert_same(lab2[b], dist2[b], *oracle.scan_topk(xx, ii, Q2[b], 10, threads=8))
    assert lab2[0][0] == n + 4 and lab2[1][0] == n + 37


def test_f6_on_topical_rows_and_a_bf16_index(dawn, oracle):
    n = 200_000
    idx = _mk(dawn, n, dist=4)
    Q = np.concatenate([synth.unit_rows_topical(1, (1 << 40) + 256 * i, 1) for i in range(24)])
    want = oracle.scan_topk_synth(1, 0, n, 1, Q, 10, dist=4)
    lab, dist, found = idx.search_batch(Q, 10)
    for b in range(len(Q)):
        _assert_same(lab[b], dist[b], want[0][b], want[1][b])
    assert idx.stats()["fallbacks"] == 0
    idh = _mk(dawn, 50_000, dtype="bf16")
    xh = synth.round_bf16(oracle.unit_rows(1, 0, 50_000))
    ids = np.arange(1, 50_001, dtype=np.uint64)
    Qh = synth.unit_rows(2, 0, 12)
    lab, dist, found = idh.search_batch(Qh, 20)
    for b in range(12):
        _assert_same(lab[b], dist[b], *oracle.scan_topk(xh, ids, Qh[b], 20))


@pytest.mark.parametrize("opts", [{"f6_refine_rows": 0}, {"f6_stagger": 8}, {"f6_stagger": 0, "f6_refine_rows": 0}, {"f6_target": 256}])
def test_f6_variants_answer_alike(dawn, oracle, opts):
    """The other forms of the filter — survivors re-scored on the int8 shadow instead of the f32 rows, the register-ring pass instead
    of the LDS-staged one, a threshold that is far too tight (most searches end in the ladder) — give the oracle's answers too."""
    n = 150_001  # (9376 tiles: the last group of 8 is ragged, the last tile holds one row)
    idx = _mk(dawn, n)
    for name, v in opts.items():
        idx.set_option(name, v)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(5, 0, 61), synth.planted_queries(1, [0, n - 1], 9)])
    lab, dist, found = idx.search_batch(Q, 10)
    for b, q in enumerate(Q):
        assert found[b] == 10
        _assert_same(lab[b], dist[b], *oracle.scan_topk(x, ids, q, 10, threads=8))
    assert lab[61][0] == 1 and lab[62][0] == n
    assert idx.stats()["fallbacks"] == 0


def test_f6_feedback_suspends_the_filter_where_it_loses(dawn, oracle):
    """An index whose FP6-filtered queries mostly end in the ladder (here: a threshold that is far too tight; in the field: topical
    rows at 100 M) hands its batches back to the int8 pass — after one window of 1024 queries above 30 %, or as soon as more than half of
    at least 256 queries ended there —, probes again later, and answers alike throughout; with "ladder_feedback" = 0 it never does."""
    n = 120_000
    idx = _mk(dawn, n)
    idx.set_option("f6_target", 256)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(6, 0, 256)
    first = idx.search_batch(Q, 10)
    for b in (0, 100, 255):
        _assert_same(first[0][b], first[1][b], *oracle.scan_topk(x, ids, Q[b], 10, threads=8))
    for _ in range(9):
        lab, dist, _ = idx.search_batch(Q, 10)
        assert np.array_equal(lab, first[0]) and np.array_equal(dist.view(np.uint32), first[1].view(np.uint32))
    s = idx.stats_f6()
    assert 1 <= s["f6_batches"] <= 4 and s["f6_suspended"] >= 4 and s["f6_batches"] + s["f6_suspended"] == 10, s
    assert idx.stats()["bounded"] >= 128
    idx.set_option("ladder_feedback", 0)
    for _ in range(8):
        idx.search_batch(Q, 10)
    s2 = idx.stats_f6()
    assert s2["f6_suspended"] == s["f6_suspended"] and s2["f6_batches"] == s["f6_batches"] + 8, (s, s2)
    # where the certificates hold it stays on
    idy = _mk(dawn, n)
    for _ in range(10):
        idy.search_batch(Q, 10)
    assert idy.stats_f6() == {"f6_batches": 10, "f6_suspended": 0}


def test_f6_behind_a_sharded_handle(dawn, oracle):
    """The option reaches every shard of a sharded handle (three logical shards on one device): each shard's batches filter on its own
    FP6 shadow, the merged answers are the oracle's, the shadows' memory shows in the handle's total and goes back."""
    n = 90_000
    sh = dawn.VectorIndex(devices=[0, 0, 0])
    sh.set_option("shard_chunk", 4096)
    sh.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(7, 0, 40), synth.planted_queries(1, [5, n - 2], 9)])
    base = sh.search_batch(Q, 10)
    m0 = sh.memory()["shadows"]
    sh.set_option("f6_min_rows", 0)
    sh.set_option("f6_shadow", 1)
    lab, dist, found = sh.search_batch(Q, 10)
    assert sh.memory()["shadows"] - m0 >= n * 288
    for b, q in enumerate(Q):
        _assert_same(lab[b], dist[b], *oracle.scan_topk(x, ids, q, 10, threads=8))
    assert np.array_equal(lab, base[0]) and np.array_equal(dist.view(np.uint32), base[1].view(np.uint32))
    assert lab[40][0] == 6 and lab[41][0] == n - 1
    sh.set_option("f6_shadow", 0)
    assert sh.memory()["shadows"] <= m0
    assert sh.stats()["fallbacks"] == 0


def test_f6_auto_policy_and_out_of_hbm_order(dawn, oracle):
    """"f6_shadow" = 2 (the default): an index of at least f6_min_rows rows builds the FP6 shadow by itself when that leaves 24 GiB of HBM free —
    and not otherwise ("f6_shadow" = 1 still does); when its allocation fails the int8 pass answers the same bits (the FP6 shadow is
    the first to go when HBM runs out: debug_fail_alloc bit 3)."""
    import torch
    n = 70_000
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 40)
    idx = dawn.VectorIndex(0)
    idx.set_option("f6_min_rows", 0)  # (default 64 Mi rows; the option's own default stays auto)
    idx.fill_synthetic(1, 0, n, 1)
    m_auto = idx.memory()["shadows"]
    b0 = idx.stats_f6()["f6_batches"]
    lab, dist, found = idx.search_batch(Q, 10)
    assert idx.stats_f6()["f6_batches"] == b0 + 1  # auto built it and the batch went through it
    for b in range(0, 40, 3):
        _assert_same(lab[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 10))
    # the allocation fails: no FP6 shadow, the int8 pass answers
    idx.set_option("debug_fail_alloc", 8)
    assert idx.memory()["shadows"] < m_auto - n * 280
    lab2, dist2, _ = idx.search_batch(Q, 10)
    assert idx.stats_f6()["f6_batches"] == b0 + 1
    assert np.array_equal(lab, lab2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32))
    idx.set_option("debug_fail_alloc", 0)
    assert idx.memory()["shadows"] == m_auto
    idx.close()
    # less than 24 GiB would stay free: auto declines, 1 insists
    free, _tot = torch.cuda.mem_get_info()
    ballast = torch.empty((max(free - (20 << 30), 1),), dtype=torch.uint8, device="cuda:0")
    try:
        idx = dawn.VectorIndex(0)
        idx.set_option("f6_min_rows", 0)
        idx.fill_synthetic(1, 0, n, 1)
        b0 = idx.stats_f6()["f6_batches"]
        lab3, dist3, _ = idx.search_batch(Q, 10)
        assert idx.stats_f6()["f6_batches"] == b0 and idx.memory()["shadows"] < m_auto - n * 280
        idx.set_option("f6_shadow", 1)
        lab4, dist4, _ = idx.search_batch(Q, 10)
        assert idx.stats_f6()["f6_batches"] == b0 + 1 and idx.memory()["shadows"] == m_auto
        for l_, d_ in ((lab3, dist3), (lab4, dist4)):
            assert np.array_equal(lab, l_) and np.array_equal(dist.view(np.uint32), d_.view(np.uint32))
        idx.close()
    finally:
        del ballast
        torch.cuda.empty_cache()
