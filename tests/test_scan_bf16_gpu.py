"""GPU parity tests for the bf16 index (configs[4] of BASELINE.json: bf16 rows, f32 accumulation).

Oracle per SURVEY §8(d): the CPU scan (oracle/dawn_oracle.c) over the bf16-ROUNDED rows — the index stores
round-to-nearest-even bf16 of the f32 vectors it is given and scores their exact f32 widening in the reference's
sequential f32 order, so distances must be bit-identical and label order identical, as for the f32 index."""
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402


@pytest.fixture(params=["i8", "own"])
def shadow(request, monkeypatch):
    """Filter source of a bf16 index: the int8 shadow of its rows (default; scan_i8.hip) or the bf16 rows themselves
    (DAWN_I8_SHADOW=0, read when an index is created)."""
    monkeypatch.setenv("DAWN_I8_SHADOW", "1" if request.param == "i8" else "0")
    return request.param


def _mk(dawn, n):
    idx = dawn.VectorIndex(0, dtype="bf16")
    idx.fill_synthetic(1, 0, n, 1)
    return idx


def _same(lab, dist, olab, odist):
    assert np.array_equal(lab, olab), (lab, olab)
    assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32)), (dist, odist)


def test_bf16_rows_are_rounded_spec_rows(dawn, oracle):
    n = 3000
    idx = _mk(dawn, n)
    rows, ids = idx.get_rows(0, n)
    want = synth.round_bf16(oracle.unit_rows(1, 0, n))
    assert np.array_equal(rows.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(ids, np.arange(1, n + 1, dtype=np.uint64))
    # the same rows through add_batch (host f32 in, rounded on the device)
    idx2 = dawn.VectorIndex(0, dtype="bf16")
    idx2.add_batch(ids, oracle.unit_rows(1, 0, n))
    idx2.add(n + 1, oracle.unit_rows(1, n, 1)[0])
    rows2, _ = idx2.get_rows(0, n + 1)
    assert np.array_equal(rows2[:n].view(np.uint32), want.view(np.uint32))
    assert np.array_equal(rows2[n].view(np.uint32), synth.round_bf16(oracle.unit_rows(1, n, 1)[0]).view(np.uint32))


@pytest.mark.parametrize("n", [1, 3, 4, 5, 63, 64, 65, 255, 1000, 4097, 100_003])
@pytest.mark.parametrize("k", [1, 10, 20])
def test_bf16_scan_matches_oracle_sizes(dawn, oracle, n, k, shadow):
    idx = _mk(dawn, n)
    x = synth.round_bf16(oracle.unit_rows(1, 0, n))
    ids = np.arange(1, n + 1, dtype=np.uint64)
    for q in synth.unit_rows(2, 0, 3):
        lab, dist = idx.search(q, k)
        olab, odist = oracle.scan_topk(x, ids, q, k)
        assert len(lab) == min(k, n)
        _same(lab, dist, olab, odist)
    assert idx.stats()["fallbacks"] == 0


@pytest.mark.parametrize("n,B,k", [(100_003, 2, 10), (100_003, 7, 20), (50_000, 9, 10), (100_003, 33, 20), (31, 16, 10),
                                   (4097, 256, 20), (8193, 200, 10), (20_001, 256, 10)])
def test_bf16_batches_match_oracle(dawn, oracle, n, B, k, shadow):
    idx = _mk(dawn, n)
    x = synth.round_bf16(oracle.unit_rows(1, 0, n))
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    Q[B // 2] = synth.planted_queries(1, [n // 3], 2)[0]
    labels, dist, found = idx.search_batch(Q, k)
    for b in range(B):
        olab, odist = oracle.scan_topk(x, ids, Q[b], k, threads=8)
        assert found[b] == min(k, n)
        _same(labels[b][:found[b]], dist[b][:found[b]], olab, odist)
    assert labels[B // 2][0] == n // 3 + 1
    assert idx.stats()["fallbacks"] == 0


def test_bf16_1m_batch1_and_256(dawn, oracle, shadow):
    n = 1_000_000
    idx = _mk(dawn, n)
    x = synth.round_bf16(oracle.unit_rows(1, 0, n))
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 256)
    labels, dist, found = idx.search_batch(Q, 10)
    for b in range(0, 256, 16):
        _same(labels[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 10, threads=8))
    for b in (0, 100, 255):  # the streaming (batch-1) path agrees with the matrix-core path
        lab, d = idx.search(Q[b], 10)
        _same(lab, d, labels[b], dist[b])
    assert idx.stats()["fallbacks"] == 0


def test_bf16_forced_exact_pass_and_gate_and_save_load(dawn, oracle, shadow):
    n = 20_000
    idx = _mk(dawn, n)
    x = synth.round_bf16(oracle.unit_rows(1, 0, n))
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 3)
    want = [oracle.scan_topk(x, ids, q, 20) for q in Q]
    idx.set_option("force_fallback", 1)
    labels, dist, found = idx.search_batch(Q, 20)
    for b in range(3):
        _same(labels[b], dist[b], *want[b])
    idx.set_option("force_fallback", 0)
    with pytest.raises(dawn.NotNormalizedError):
        idx.add_batch(np.array([7, 8], dtype=np.uint64), np.stack([Q[0], Q[1] * 1.5]))
    assert idx.size() == n
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "index.dawn")
        idx.save(path)
        idx2 = dawn.VectorIndex(0, dtype="bf16")
        idx2.load(path)
        assert idx2.size() == n
        lab, dd = idx2.search(Q[1], 20)
        _same(lab, dd, *want[1])


def _hard_rows(dawn):
    """Random unit rows, near-duplicates of a query and sparse rows with tiny components."""
    rng = np.random.default_rng(11)
    base = synth.unit_rows(1, 0, 5000)
    sparse = np.zeros((600, 384), dtype=np.float32)
    for i in range(600):
        sparse[i, rng.integers(0, 384, 3)] = rng.standard_normal(3).astype(np.float32)
        sparse[i] += (1e-6 * rng.standard_normal(384)).astype(np.float32)
        sparse[i] = dawn.normalize(sparse[i])
    return np.concatenate([base, sparse])


def test_bf16_filter_errors_within_their_bounds(dawn):
    """The certificates of a bf16 index assume |filter score - exact dot over the STORED (bf16) rows| <=
    FILTER_EPS_BF16_STREAM = 7e-5 for the streaming filter (query as hi + lo bf16 columns) and <= FILTER_EPS_BF16_MFMA =
    4.1e-3 for the matrix-core pass (one bf16 image of the query): measured against float64."""
    rows = _hard_rows(dawn)
    idx = dawn.VectorIndex(0, dtype="bf16")
    idx.add_batch(np.arange(1, len(rows) + 1, dtype=np.uint64), rows)
    stored = synth.round_bf16(rows).astype(np.float64)
    Q = np.concatenate([synth.unit_rows(2, 0, 24), synth.planted_queries(1, [5, 77, 4000], 3), rows[5000:5005]])
    # the int8 shadow of the bf16 rows (default): every filter score bounds the exact dot of the STORED row from above
    ub = idx.debug_filter_scores(Q)
    slack = ub.astype(np.float64) - Q.astype(np.float64) @ stored.T
    assert slack.min() > -4e-6 and np.median(slack) < 0.02 and slack.max() < 0.03, (slack.min(), np.median(slack), slack.max())
    for q in Q[:4]:
        sc, rr = idx.debug_stream_lists(q)
        valid = rr != 0xFFFFFFFF
        assert (sc[valid].astype(np.float64) - stored[rr[valid].astype(np.int64)] @ q.astype(np.float64)).min() > -4e-6
    idx.set_option("i8_shadow", 0)
    # matrix-core pass on the bf16 rows themselves: dense scores of every row
    f = idx.debug_filter_scores(Q)
    exact = Q.astype(np.float64) @ stored.T
    assert f.shape == exact.shape
    err = np.abs(f.astype(np.float64) - exact).max()
    assert err < 4.1e-3 / 2, err
    # streaming filter: whatever it lists carries the hi+lo score
    worst = 0.0
    for q in Q[:8]:
        sc, rr = idx.debug_stream_lists(q)
        valid = rr != 0xFFFFFFFF
        got = rr[valid].astype(np.int64)
        assert len(np.unique(got)) == len(got)
        ex = stored[got] @ q.astype(np.float64)
        worst = max(worst, np.abs(sc[valid].astype(np.float64) - ex).max())
        # ... and the lists hold every row that clears the 64th best by twice the bound
        allx = stored @ q.astype(np.float64)
        order = np.argsort(-allx, kind="stable")
        need = order[:64][allx[order[:64]] > allx[order[63]] + 2 * 7e-5]
        assert set(need.tolist()) <= set(got.tolist())
    assert worst < 7e-5 / 2, worst
