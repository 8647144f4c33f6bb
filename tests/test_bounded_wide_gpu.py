"""GPU parity tests of the WIDE batch form of the bounded exact pass (dawnsearch_amd/csrc/scan_bounded.hip:
scan_bounded_i8_wide_kernel + bounded_wide_finish_kernel): 64 flagged queries of a batch per stream of the int8 shadow, no
lists — (row, query) pairs past the int8 bound are re-tested on the f32 row, scored in the reference's order
(src/search/vector.rs:128-134) when they can still matter, and appended; the finish kernel sorts by (distance, row).  Bar as
everywhere: labels and distance BITS of the CPU oracle; dawn_index_debug_raw_stats proves which form answered ([0]: the wide
form, [4]: the bounded pass as a whole, [1]: exact passes over all rows = 0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402

QROW0 = 1 << 40


def _same(lab, dist, olab, odist):
    assert len(lab) == len(olab)
    assert np.array_equal(lab, olab), (lab, olab)
    assert np.array_equal(np.asarray(dist).view(np.uint32), np.asarray(odist).view(np.uint32)), (dist, odist)


@pytest.mark.parametrize("n", [1, 33, 4097, 150_001])
@pytest.mark.parametrize("B", [2, 37, 64, 65, 256])
@pytest.mark.parametrize("k", [1, 10, 64])
def test_forced_ladder_batches_take_the_wide_form(dawn, oracle, n, B, k):
    """force_fallback = 2: every certificate of the batch fails; all B queries go through the wide form (one, two or four groups
    of up to 64), ragged groups and n < k included."""
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    Q[B // 2] = synth.planted_queries(1, [n // 2], 4)[0]
    idx.set_option("force_fallback", 2)
    r0 = idx.stats_raw()
    lab, dist, found = idx.search_batch(Q, k)
    r1 = idx.stats_raw()
    for b in (range(B) if B <= 37 else list(range(0, B, 7)) + [B // 2, B - 1]):
        assert found[b] == min(k, n)
        _same(lab[b][:found[b]], dist[b][:found[b]], *oracle.scan_topk(x, ids, Q[b], k))
    assert r1[4] - r0[4] == B and r1[1] == r0[1], (r0, r1)
    # every one of them by the wide form (an index below 4096 rows: by the 16-query form — with k >= n every row of every query is a result)
    assert r1[0] - r0[0] == (B if n >= 4096 else 0), (r0, r1)
    # ... and the same answers with the wide form switched off (sixteen per stream)
    idx.set_option("bounded_wide", 0)
    lab2, dist2, found2 = idx.search_batch(Q, k)
    r2 = idx.stats_raw()
    assert np.array_equal(lab, lab2) and np.array_equal(dist.view(np.uint32), dist2.view(np.uint32)) and np.array_equal(found, found2)
    assert r2[0] == r1[0] and r2[4] - r1[4] == B


def test_a_wave_out_of_room_leaves_its_queries_to_the_sixteen_query_form(dawn, oracle):
    """4200 identical rows: every query's k-th distance IS the tie, all 4200 rows pass both bounds, and a wave that holds 32 of them
    collects 64 queries x 32 rows = 2048 pairs, more than its queue (1024): the queries it runs out of room for keep their flag
    and the 16-query form answers them — first-inserted rows first (src/search/best_results.rs:56)."""
    n = 4200
    v = synth.unit_rows(7, 0, 1)[0]
    idx = dawn.VectorIndex(0)
    idx.add_batch(np.arange(1, n + 1, dtype=np.uint64), np.repeat(v[None, :], n, axis=0))
    x = np.repeat(v[None, :], n, axis=0)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 64)
    idx.set_option("force_fallback", 2)
    r0 = idx.stats_raw()
    lab, dist, found = idx.search_batch(Q, 10)
    r1 = idx.stats_raw()
    for b in range(0, 64, 5):
        _same(lab[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 10))
    assert r1[4] - r0[4] == 64 and r1[1] == r0[1] and r1[0] - r0[0] < 64, (r0, r1)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("dist_id", [4, 5])
def test_topical_batches_wide_equals_oracle(dawn, oracle, dtype, dist_id):
    """400 k topical rows with the candidate buffers shrunk until certificates fail as they do on 100 M rows (tests/test_ladder_gpu.py):
    a batch of 96 queries, half of them inside the three largest clusters."""
    n = 400_000
    idx = dawn.VectorIndex(0, dtype=dtype)
    idx.set_option("synth_dist", dist_id)
    idx.fill_synthetic(1, 0, n, 1)
    idx.set_option("mfma_target", 64)
    runs = dist_id == 5
    qrows = []
    i = 0
    while len(qrows) < 48:
        r = QROW0 + i * 256
        i += 1
        if int(synth.topical_cluster(1, np.array([r]), runs=runs)[0][0]) in (0, 1, 2):
            qrows.append(r)
    qrows += [QROW0 + 256 * (1000 + j) for j in range(48)]
    Q = np.stack([synth.unit_rows_topical(1, r, 1, runs=runs)[0] for r in qrows])
    for k in (10, 20):
        want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=dist_id, bf16=(dtype == "bf16"))
        r0 = idx.stats_raw()
        lab, dist, found = idx.search_batch(Q, k)
        r1 = idx.stats_raw()
        for b in range(len(Q)):
            assert found[b] == k
            _same(lab[b], dist[b], want[0][b], want[1][b])
        assert r1[1] == r0[1], (r0, r1)
        assert r1[4] - r0[4] >= 10 and r1[0] - r0[0] == r1[4] - r0[4], (r0, r1)  # the ladder ran, all of it in the wide form


@pytest.mark.parametrize("copies", [700, 3000])
def test_ties_and_overflow(dawn, oracle, copies):
    """`copies` identical rows: a query next to them ties `copies` rows at its best distance — first-inserted wins
    (src/search/best_results.rs:56).  700 copies: the wide form sorts the ties by row.  3000 > BOUNDED_WIDE_CAP = 2048: the query's
    buffer overflows, it keeps its flag and the 16-query form (lists) answers it; its neighbours in the batch stay with the wide form."""
    n = 60_000
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    v = synth.unit_rows(7, 0, 1)[0]
    idx.add_batch(np.arange(n + 1, n + copies + 1, dtype=np.uint64), np.repeat(v[None, :], copies, axis=0))
    idx.add_batch(np.arange(n + copies + 1, n + copies + 501, dtype=np.uint64), synth.unit_rows(8, 0, 500))
    x = np.concatenate([oracle.unit_rows(1, 0, n), np.repeat(v[None, :], copies, axis=0), synth.unit_rows(8, 0, 500)])
    ids = np.arange(1, len(x) + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 40)
    noisy = v + 0.01 * synth.unit_rows(9, 0, 1)[0]
    Q[3] = v
    Q[17] = noisy / np.float32(np.sqrt(np.sum(noisy.astype(np.float64) ** 2)))
    idx.set_option("force_fallback", 2)
    for k in (10, 64):
        r0 = idx.stats_raw()
        lab, dist, found = idx.search_batch(Q, k)
        r1 = idx.stats_raw()
        for b in range(len(Q)):
            _same(lab[b][:found[b]], dist[b][:found[b]], *oracle.scan_topk(x, ids, Q[b], k))
        assert list(lab[3][:3]) == [n + 1, n + 2, n + 3]
        assert r1[4] - r0[4] == len(Q) and r1[1] == r0[1]
        assert r1[0] - r0[0] == (len(Q) if copies <= 2048 else len(Q) - 2), (r0, r1)


def test_wide_form_on_a_sharded_handle(dawn, oracle):
    """Two logical shards behind one handle: each shard's ladder runs its own wide form; merged answers = the oracle's."""
    n = 50_000
    sh = dawn.VectorIndex(devices=[0, 0])
    sh.set_option("shard_chunk", 1024)
    sh.fill_synthetic(1, 0, n, 1)
    sh.set_option("force_fallback", 2)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 70)
    lab, dist, found = sh.search_batch(Q, 10)
    for b in range(0, 70, 3):
        _same(lab[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 10))
    r = sh.stats_raw()
    assert r[0] == 140 and r[4] == 140 and r[1] == 0, r
    fb = sh.stats_batch_feedback()  # (a sharded handle reports the sums over its shards: ADVICE r4)
    assert fb["f6_batches"] == 0 and fb["deepened_batches"] == 0 and sh.stats_f6()["f6_suspended"] == 0
