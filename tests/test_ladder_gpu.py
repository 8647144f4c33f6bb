"""GPU parity tests of the ladder behind a failed certificate, on the data that makes certificates fail.

Every filter of the library ends in a certificate; on isotropic synthetic rows it always holds.  On TOPICAL rows — Zipf-sized
clusters, cosine 0.5 .. 0.95 inside a cluster (dawnsearch_amd/synth.py: unit_rows_topical = option "synth_dist" 4; 5: runs of
256 consecutive rows per cluster, as one site's pages arrive back to back: src/index/warc.rs:75-86,
src/search/search_provider.rs:250-286) — a query inside a large cluster has more rows within the filter's slack of its k-th
score than the fixed-size lists hold.  Round 3 answered those from the exact pass over all rows; now the bounded exact pass on
the int8 shadow does (dawnsearch_amd/csrc/scan_bounded.hip).  Bar as everywhere: labels and distance BITS of the CPU oracle
(oracle/dawn_oracle.c: src/search/vector.rs:128-134 + exact top-k), with dawn_index_stats* proving which rung answered.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402

QROW0 = 1 << 40  # queries: FURTHER rows of the same stream (new pages on the same topics)


def _same(lab, dist, olab, odist):
    assert len(lab) == len(olab)
    assert np.array_equal(lab, olab), (lab, olab)
    assert np.array_equal(np.asarray(dist).view(np.uint32), np.asarray(odist).view(np.uint32)), (dist, odist)


def _topical_index(dawn, n, dist, packed=True, dtype="f32"):
    idx = dawn.VectorIndex(0, dtype=dtype)
    idx.set_option("synth_dist", dist)
    if packed:
        idx.set_option("i6_min_rows", 0)
    idx.fill_synthetic(1, 0, n, 1)
    return idx


def _topical_queries(dist, nq, clusters=None):
    """Rows QROW0 + i * 256 of the stream (a different run each, for dist 5).  clusters: keep only queries of these clusters."""
    out = []
    i = 0
    while len(out) < nq:
        r = QROW0 + i * 256
        i += 1
        if clusters is not None and int(synth.topical_cluster(1, np.array([r]), runs=(dist == 5))[0][0]) not in clusters:
            continue
        out.append(synth.unit_rows_topical(1, r, 1, runs=(dist == 5))[0])
    return np.stack(out)


@pytest.mark.parametrize("dist", [4, 5])
def test_topical_generator_matches_numpy_and_oracle(dawn, oracle, dist):
    n = 3000
    idx = _topical_index(dawn, n, dist, packed=False)
    rows, ids = idx.get_rows(0, n)
    assert np.array_equal(rows.view(np.uint32), synth.unit_rows_topical(1, 0, n, runs=(dist == 5)).view(np.uint32))
    assert np.array_equal(rows.view(np.uint32), oracle.unit_rows_topical(1, 0, n, runs=(dist == 5)).view(np.uint32))
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    qi.fill_synthetic(1, QROW0 + 5, 70, 1)
    got = qi.get_rows(0, 70)[0]
    assert np.array_equal(got.view(np.uint32), synth.unit_rows_topical(1, QROW0 + 5, 70, runs=(dist == 5)).view(np.uint32))
    assert np.all(np.abs(np.linalg.norm(rows.astype(np.float64), axis=1) - 1.0) < 1e-6)


@pytest.mark.parametrize("dist", [4, 5])
@pytest.mark.parametrize("k", [10, 20])
@pytest.mark.parametrize("bounded_packed", [0, 2])
def test_topical_index_every_rung_equals_the_oracle(dawn, oracle, dist, k, bounded_packed):
    """400 k topical rows, queries inside the three largest clusters (33 k, 17 k, 17 k rows) and anywhere.  The lists are sized
    for 100 M rows; here the grids are shrunk until they are as short of this index's clusters as the full grids are of a
    100 M-row index's (tools/clustered_probe.py: 28 % of such queries fail on 12.5 M rows): packed stream 4 workgroups x 8 waves x
    8 entries, int8 stream 2 workgroups, 64 candidates per query in a batch.  Packed stream, int8 stream, a batch — all bit-equal
    to the oracle's scan of the same rows, no exact pass anywhere, the bounded pass counted where certificates failed.
    bounded_packed: the bounded pass of a single query streams the int8 shadow (0) or the packed 5-bit shadow (2: forced; by
    default wherever the packed shadow is live, from 2 Mi rows)."""
    n = 400_000
    idx = _topical_index(dawn, n, dist)
    idx.set_option("bounded_packed", bounded_packed)
    idx.set_option("i6_scan_blocks", 4)
    idx.set_option("i6_refine", 8)
    idx.set_option("shadow_scan_blocks", 2)
    idx.set_option("mfma_target", 64)
    Q = np.concatenate([_topical_queries(dist, 6, clusters={0, 1, 2}), _topical_queries(dist, 6)])
    want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=dist)
    # one query at a time on the packed shadow
    for b, q in enumerate(Q):
        _same(*idx.search(q, k), want[0][b], want[1][b])
    st = idx.stats()
    assert st["fallbacks"] == 0 and st["bounded"] >= 3, st
    # ... on the int8 shadow
    idx.set_option("i6_shadow", 0)
    for b, q in enumerate(Q):
        _same(*idx.search(q, k), want[0][b], want[1][b])
    st2 = idx.stats()
    assert st2["fallbacks"] == 0 and st2["bounded"] > st["bounded"], st2
    # ... and as one batch (matrix-core pass; its flagged queries share one stream of the bounded pass)
    lab, dist_, found = idx.search_batch(Q, k)
    for b in range(len(Q)):
        assert found[b] == k
        _same(lab[b], dist_[b], want[0][b], want[1][b])
    st3 = idx.stats()
    assert st3["fallbacks"] == 0 and st3["bounded"] > st2["bounded"], st3
    # with the rung switched off the exact pass answers the same (round 3's behaviour)
    idx.set_option("i6_shadow", 1)
    idx.set_option("bounded_pass", 0)
    for b in (0, 1, 7):
        _same(*idx.search(Q[b], k), want[0][b], want[1][b])
    st4 = idx.stats()
    assert st4["bounded"] == st3["bounded"] and st4["fallbacks"] >= 1


@pytest.mark.parametrize("n", [1, 31, 64, 65, 1000, 4097, 100_003])
@pytest.mark.parametrize("k", [1, 10, 64])
def test_forced_ladder_on_uniform_rows_sizes(dawn, oracle, n, k):
    """force_fallback = 2: every certificate is made to fail and the ladder answers — the bounded pass for everything the int8
    shadow covers.  Sizes around the sub-tile / list boundaries, k up to the list length."""
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 2), synth.planted_queries(1, [n // 2], 4)])
    idx.set_option("force_fallback", 2)
    for q in Q:
        lab, dist = idx.search(q, k)
        assert len(lab) == min(k, n)
        _same(lab, dist, *oracle.scan_topk(x, ids, q, k))
    lab, dist, found = idx.search_batch(Q, k)
    for b, q in enumerate(Q):
        _same(lab[b][:found[b]], dist[b][:found[b]], *oracle.scan_topk(x, ids, q, k))
    st = idx.stats()
    assert st["bounded"] == 6 and st["fallbacks"] == 0, st


def test_forced_ladder_batch_of_256_and_groups_of_sixteen(dawn, oracle):
    """A whole batch through the bounded pass: 256 flagged queries = 16 streams of 16 queries; 37 queries: two full groups and a
    ragged one."""
    n = 200_000
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 256)
    Q[5] = synth.planted_queries(1, [777], 4)[0]
    idx.set_option("force_fallback", 2)
    for nb in (256, 37):
        s0 = idx.stats()
        lab, dist, found = idx.search_batch(Q[:nb], 10)
        s1 = idx.stats()
        assert s1["bounded"] - s0["bounded"] == nb and s1["fallbacks"] == 0
        for b in list(range(0, nb, 9)) + [5, nb - 1]:
            _same(lab[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 10, threads=8))
    assert lab[5][0] == 778


def test_duplicates_beyond_every_list_take_the_bounded_pass(dawn, oracle):
    """20 000 copies of the best row (the KAT of SURVEY 8c at a size no list holds): the bounded pass scores all of them exactly
    and returns the earliest-added ones; a bf16 index likewise."""
    base = synth.unit_rows(1, 0, 5000)
    q = synth.planted_queries(1, [7], 3)[0]
    rows = np.concatenate([base, np.repeat(base[7:8], 20_000, axis=0), base[:100]])
    ids = np.arange(1000, 1000 + len(rows), dtype=np.uint64)
    for dtype in ("f32", "bf16"):
        ref = rows if dtype == "f32" else synth.round_bf16(rows)
        for packed in (True, False):
            idx = dawn.VectorIndex(0, dtype=dtype)
            if packed:
                idx.set_option("i6_min_rows", 0)
            idx.add_batch(ids, rows)
            for k in (20, 64):
                _same(*idx.search(q, k), *oracle.scan_topk(ref, ids, q, k))
            lab, dist, found = idx.search_batch(np.stack([q, base[3], q]), 20)
            for b, qq in enumerate((q, base[3], q)):
                _same(lab[b], dist[b], *oracle.scan_topk(ref, ids, qq, 20))
            st = idx.stats()
            assert st["fallbacks"] == 0 and st["bounded"] >= 4, (dtype, packed, st)


def test_bounded_pass_behind_a_sharded_handle(dawn, oracle):
    n = 300_000
    sh = dawn.VectorIndex(devices=[0, 0, 0])
    sh.set_option("shard_chunk", 4096)
    sh.set_option("synth_dist", 4)
    sh.set_option("i6_min_rows", 0)
    sh.fill_synthetic(1, 0, n, 1)
    Q = _topical_queries(4, 4, clusters={0, 1})
    want = oracle.scan_topk_synth(1, 0, n, 1, Q, 10, dist=4)
    sh.set_option("force_fallback", 2)  # every shard's certificate fails: every shard's bounded pass answers
    for b, q in enumerate(Q):
        _same(*sh.search(q, 10), want[0][b], want[1][b])
    lab, dist, found = sh.search_batch(Q, 10)
    for b in range(len(Q)):
        _same(lab[b], dist[b], want[0][b], want[1][b])
    st = sh.stats()
    assert st["fallbacks"] == 0 and st["bounded"] == 3 * 8, st


def test_without_the_int8_shadow_the_exact_pass_still_answers(dawn, oracle, monkeypatch):
    """The bounded pass reads the int8 shadow; an index that does not keep one (i8_shadow = 0: f16 shadow) falls back as before."""
    monkeypatch.setenv("DAWN_I8_SHADOW", "0")
    n = 100_000
    idx = _topical_index(dawn, n, 4, packed=False)
    Q = _topical_queries(4, 3, clusters={0})
    want = oracle.scan_topk_synth(1, 0, n, 1, Q, 10, dist=4)
    idx.set_option("force_fallback", 2)
    for b, q in enumerate(Q):
        _same(*idx.search(q, 10), want[0][b], want[1][b])
    st = idx.stats()
    assert st["bounded"] == 0 and st["fallbacks"] == 3


def test_ladder_feedback_demotes_and_recovers(dawn, oracle):
    """More than a third of the packed certificates failing over a window of 32 single queries: the index sends the next ones to
    the bounded pass directly (one 384-B stream instead of a 240-B stream and then the 384-B one), probes the packed stream again
    afterwards, starts afresh after a mutation, and never demotes queries that certify.  Same answers throughout."""
    n = 200_000
    idx = _topical_index(dawn, n, 4)
    idx.set_option("i6_scan_blocks", 4)  # (lists as short of this index's clusters as the full grid is of a 100 M-row index's)
    idx.set_option("i6_refine", 8)
    Qbad = _topical_queries(4, 8, clusters={0, 1, 2})
    want = oracle.scan_topk_synth(1, 0, n, 1, Qbad, 10, dist=4)
    for it in range(12):  # 96 searches: window 1 (32 packed, all failing) -> demoted
        for b, q in enumerate(Qbad):
            _same(*idx.search(q, 10), want[0][b], want[1][b])
    st = idx.stats()
    assert st["fallbacks"] == 0 and st["bounded"] == 96 and st["packed_failures"] == 32 and st["demoted"] == 64, st
    # a trickle of adds (the reference inserts a page between searches: src/search/search_service.rs:158-171) does NOT start the feedback
    # over — the index stays demoted; an option change (or growth by an eighth, a load, a clear) does: the packed stream is tried again
    extra = synth.unit_rows(9, 0, 1)
    idx.add(n + 1, extra[0])
    grown = (np.concatenate([oracle.unit_rows_topical(1, 0, n), extra]), np.arange(1, n + 2, dtype=np.uint64))
    _same(*idx.search(Qbad[0], 10), *oracle.scan_topk(*grown, Qbad[0], 10, threads=8))
    st2 = idx.stats()
    assert st2["packed_failures"] == 32 and st2["demoted"] == 65, st2
    idx.set_option("i6_shadow", 1)  # (an option that re-prepares the searches)
    _same(*idx.search(Qbad[0], 10), *oracle.scan_topk(*grown, Qbad[0], 10, threads=8))
    st2 = idx.stats()
    assert st2["packed_failures"] == 33 and st2["demoted"] == 65, st2
    # feedback off: every search tries the packed stream first
    idx.set_option("ladder_feedback", 0)
    for it in range(6):
        for q in Qbad:
            idx.search(q, 10)
    st3 = idx.stats()
    assert st3["demoted"] == 65 and st3["packed_failures"] == 33 + 48
    # queries that certify are never demoted
    idx2 = dawn.VectorIndex(0)
    idx2.set_option("i6_min_rows", 0)
    idx2.fill_synthetic(1, 0, n, 1)
    Q = synth.unit_rows(2, 0, 8)
    for it in range(10):
        for q in Q:
            idx2.search(q, 10)
    st = idx2.stats()
    assert st["demoted"] == 0 and st["packed_failures"] == 0 and st["bounded"] == 0 and st["fallbacks"] == 0


def test_bounded_pass_notices_an_impossible_threshold(dawn, oracle):
    """The one assumption of the bounded pass — the threshold it starts from bounds the final k-th distance from above — is
    checked by its last workgroup; no producer of the library hands over a threshold that fails it, so a test hook does: the
    workgroup then scans all rows exactly by itself (slow, correct) and the query counts as a fallback."""
    n = 60_000
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 2), synth.planted_queries(1, [n // 3], 4)])
    idx.set_option("ladder_feedback", 2)  # every single query: the bounded pass alone, started without a threshold
    for q in Q:
        _same(*idx.search(q, 10), *oracle.scan_topk(x, ids, q, 10))
    st = idx.stats()
    assert st["demoted"] == 3 and st["bounded"] == 3 and st["fallbacks"] == 0 and st["packed_failures"] == 0, st
    idx.set_option("debug_bad_threshold", 1)
    for q in Q:
        _same(*idx.search(q, 20), *oracle.scan_topk(x, ids, q, 20))
    st = idx.stats()
    assert st["demoted"] == 6 and st["bounded"] == 3 and st["fallbacks"] == 3, st


def test_batch_feedback_deepens_the_thresholds_of_a_ladder_heavy_index(dawn, oracle):
    """Batches on topical rows whose queries sit in the largest clusters end in the ladder often; after one window of 1024 batched
    queries above 10 % the index's batches TRY four times as deep ("mfma_target" 4096 instead of 1024) for a window, and keep that
    depth only if it saves the ladder a whole stream of its wide form (64 flagged queries per stream: ceil(flagged / 64) per batch);
    the answers stay the oracle's; an index of well-spread rows never deepens; "ladder_feedback" = 0 switches it off."""
    import math
    n = 400_000
    idx = _topical_index(dawn, n, 4, packed=False)
    Q = np.concatenate([_topical_queries(4, 192, clusters={0, 1, 2}), _topical_queries(4, 64)])
    want = oracle.scan_topk_synth(1, 0, n, 1, Q[::16], 10, dist=4)
    rates = []
    for it in range(12):
        s0 = idx.stats()
        lab, dist, found = idx.search_batch(Q, 10)
        s1 = idx.stats()
        rates.append((s1["bounded"] - s0["bounded"]) / 256.0)
        for j, b in enumerate(range(0, 256, 16)):
            _same(lab[b], dist[b], want[0][j], want[1][j])
        if it == 7:
            fbk = idx.stats_batch_feedback()
            assert fbk["deepened_batches"] == 4, (fbk, rates)  # batches 5 .. 8: the trial window behind the first window of 4 x 256 queries
    assert rates[0] > 0.10, rates                     # (the premise: this index is ladder-heavy)
    assert min(rates[4:8]) <= max(rates[:4]), rates
    streams = lambda r: math.ceil(sum(r) / len(r) * 256 / 64.0 - 0.02)
    kept = streams(rates[4:8]) < streams(rates[:4])
    assert idx.stats_batch_feedback()["deepened_batches"] == (8 if kept else 4), (rates, kept)
    assert idx.stats()["fallbacks"] == 0
    deep_now = idx.stats_batch_feedback()["deepened_batches"]
    idx.set_option("ladder_feedback", 0)
    for _ in range(6):
        idx.search_batch(Q, 10)
    assert idx.stats_batch_feedback()["deepened_batches"] == deep_now
    flat = dawn.VectorIndex(0)
    flat.fill_synthetic(1, 0, 300_000, 1)
    Qf = synth.unit_rows(2, 0, 256)
    for _ in range(8):
        flat.search_batch(Qf, 10)
    assert flat.stats_batch_feedback()["deepened_batches"] == 0 and flat.stats_batch_feedback()["rerun_answers"] == 0


@pytest.mark.parametrize("dist", [4, 5])
@pytest.mark.parametrize("k", [10, 20])
def test_batch_rerun_settles_flagged_queries(dawn, oracle, dist, k):
    """The second pass of a batch's flagged queries (option "batch_rerun" = 2: every batch; by default only on a ladder-heavy
    index): thresholds from the queries' own k-th exact distances, the same tail over ALL rows above them.  With the sampled
    thresholds made far too shallow ("mfma_target" 64 on 400 k topical rows) most certificates fail at first; the second pass
    settles those whose candidates fit, the bounded pass the rest — the oracle's answers, no exact pass."""
    n = 400_000
    idx = _topical_index(dawn, n, dist, packed=False)
    idx.set_option("mfma_target", 64)
    Q = np.concatenate([_topical_queries(dist, 40, clusters={0, 1, 2}), _topical_queries(dist, 24)])
    want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=dist)
    base = idx.search_batch(Q, k)
    b0 = idx.stats()["bounded"]
    assert b0 >= 8  # (the premise)
    idx.set_option("batch_rerun", 2)
    lab, dist_, found = idx.search_batch(Q, k)
    for b in range(len(Q)):
        assert found[b] == k
        _same(lab[b], dist_[b], want[0][b], want[1][b])
        _same(base[0][b], base[1][b], want[0][b], want[1][b])
    st = idx.stats()
    fb = idx.stats_batch_feedback()
    assert st["fallbacks"] == 0 and fb["rerun_answers"] >= 4, (st, fb)
    assert fb["rerun_answers"] + (st["bounded"] - b0) == b0, (st, fb, b0)  # (the same queries failed at first both times)


@pytest.mark.parametrize("n", [1, 31, 33, 64, 1000, 4097, 100_003])
@pytest.mark.parametrize("k", [1, 10, 64])
def test_bounded_pass_on_the_packed_shadow_sizes(dawn, oracle, n, k):
    """The bounded pass of a single query on the packed 5-bit shadow (option "bounded_packed" = 2: by default only from 2 Mi rows):
    behind certificates that are made to fail (force_fallback = 2) and as the whole search of a demoted index (ladder_feedback =
    2: no first threshold), sizes around the sub-tile boundaries, k up to the list length, an f32 and a bf16 index."""
    for dtype in ("f32", "bf16"):
        idx = dawn.VectorIndex(0, dtype=dtype)
        idx.set_option("i6_min_rows", 0)
        idx.set_option("bounded_packed", 2)
        idx.fill_synthetic(1, 0, n, 1)
        x = oracle.unit_rows(1, 0, n)
        if dtype == "bf16":
            x = synth.round_bf16(x)
        ids = np.arange(1, n + 1, dtype=np.uint64)
        Q = np.concatenate([synth.unit_rows(2, 0, 2), synth.planted_queries(1, [n // 2], 4)])
        idx.set_option("force_fallback", 2)
        for q in Q:
            lab, dist = idx.search(q, k)
            assert len(lab) == min(k, n)
            _same(lab, dist, *oracle.scan_topk(x, ids, q, k))
        idx.set_option("force_fallback", 0)
        idx.set_option("ladder_feedback", 2)
        for q in Q:
            _same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k))
        st = idx.stats()
        assert st["bounded"] == 6 and st["fallbacks"] == 0 and st["demoted"] == 3, st
        if n >= 32 * 1024:
            # ... seeded: the packed stream over the first 1/32 of the rows hands the pass its first threshold (by default from 2 Mi rows)
            idx.set_option("bounded_seed", 2)
            for q in Q:
                _same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k))
            st = idx.stats()
            assert st["bounded"] == 9 and st["fallbacks"] == 0 and st["demoted"] == 6, st


@pytest.mark.parametrize("dist", [4, 5])
def test_seeded_bounded_pass_on_topical_rows(dawn, oracle, dist):
    """A demoted index on topical rows, the bounded pass on the packed shadow seeded by a packed-stream search over the first 1/32 of
    the rows (whose own certificate may fail — with the shrunken lists it does: its k-th distance is a valid threshold all the
    same): the oracle's answers, no exact pass."""
    n = 400_000
    idx = _topical_index(dawn, n, dist)
    idx.set_option("bounded_packed", 2)
    idx.set_option("bounded_seed", 2)
    idx.set_option("ladder_feedback", 2)
    idx.set_option("i6_scan_blocks", 4)
    idx.set_option("i6_refine", 8)
    Q = np.concatenate([_topical_queries(dist, 6, clusters={0, 1, 2}), _topical_queries(dist, 6)])
    for k in (10, 20):
        want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=dist)
        for b, q in enumerate(Q):
            _same(*idx.search(q, k), want[0][b], want[1][b])
    st = idx.stats()
    assert st["fallbacks"] == 0 and st["bounded"] == 24 and st["demoted"] == 24, st


@pytest.mark.parametrize("feedback", [0, 1, 2])
def test_search_captured_into_a_hipgraph_on_a_topical_index(dawn, oracle, feedback):
    """dawn_index_search_device is launches only, every rung of the ladder predicated on the device: captured into a hipGraph and
    replayed 200 times with changing queries (certifying ones and ones that need the bounded pass), every replay returns the
    oracle's bits.  The host's only decision — filter stream first, or the bounded pass directly — is frozen at capture
    (include/dawn_hip.h): with "ladder_feedback" = 0 / 2 it is chosen explicitly, with 1 it is whatever the feedback said then;
    dawn_index_stats_ladder shows which rungs ran."""
    import torch
    n = 200_000
    idx = _topical_index(dawn, n, 4)
    idx.set_option("i6_scan_blocks", 4)  # (lists as short of this index's clusters as the full grid is of a 100 M-row index's)
    idx.set_option("i6_refine", 8)
    idx.set_option("ladder_feedback", feedback)
    Q = np.concatenate([_topical_queries(4, 6, clusters={0, 1, 2}), _topical_queries(4, 6), synth.unit_rows(2, 0, 4)])
    k = 10
    want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=4)
    dev = torch.device("cuda", 0)
    d_q = torch.zeros((384,), dtype=torch.float32, device=dev)
    blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
    p = blob.data_ptr()
    side = torch.cuda.Stream(device=dev)
    d_q.copy_(torch.from_numpy(Q[0]))
    with torch.cuda.stream(side):  # warm-up on the capture stream
        idx.search_device(d_q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, side.cuda_stream)
    side.synchronize()
    s0 = idx.stats()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        idx.search_device(d_q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, torch.cuda.current_stream().cuda_stream)
    for it in range(200):
        b = it % len(Q)
        d_q.copy_(torch.from_numpy(Q[b]))
        g.replay()
        torch.cuda.synchronize()
        raw = blob.cpu().numpy()
        lab = raw[:k * 8].view(np.uint64)
        dist = raw[k * 8:k * 12].view(np.float32)
        _same(lab, dist, want[0][b], want[1][b])
    s1 = idx.stats()
    assert s1["fallbacks"] == s0["fallbacks"]
    assert s1["bounded"] - s0["bounded"] >= (200 if feedback == 2 else 40), (s0, s1)  # the ladder's rungs ran inside the replays
    if feedback == 0:
        assert s1["packed_failures"] - s0["packed_failures"] >= 40
