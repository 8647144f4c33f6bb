"""GPU parity tests for the cosine scan + top-k (configs[1] of BASELINE.json and its edge cases).

The HIP path (through the C ABI) is compared with the CPU oracle (oracle/dawn_oracle.c, a restatement of
src/search/vector.rs:128-134 + exact top-k) on identical seeded inputs.  Bar: BIT-EXACT distances and
identical label order — the library rescoring is done in the reference's summation order, so no
tolerance is needed (north_star allows 1e-5; we assert 0).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402


@pytest.fixture(params=["i8", "f16"])
def shadow(request, monkeypatch):
    """Filter shadow of an f32 index: int8 upper-bound scores (default) or scaled f16 (DAWN_I8_SHADOW=0: the default
    of the "i8_shadow" option, read when an index is created)."""
    monkeypatch.setenv("DAWN_I8_SHADOW", "1" if request.param == "i8" else "0")
    return request.param


def _mk_index(dawn, n, seed=1, first_id=1):
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(seed, 0, n, first_id)
    return idx


def _assert_same(lab, dist, olab, odist):
    assert len(lab) == len(olab)
    assert np.array_equal(lab, olab), (lab, olab)
    assert np.array_equal(dist.view(np.uint32), odist.view(np.uint32)), (dist, odist)


def test_generator_matches_oracle(dawn, oracle):
    n = 3000
    idx = _mk_index(dawn, n)
    rows, ids = idx.get_rows(0, n)
    ref = oracle.unit_rows(1, 0, n)
    assert np.array_equal(rows.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(ids, np.arange(1, n + 1, dtype=np.uint64))
    assert np.array_equal(rows.view(np.uint32), synth.unit_rows(1, 0, n).view(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 127, 1000, 4097, 100_003])
@pytest.mark.parametrize("k", [1, 10, 20, 64])
def test_scan_matches_oracle_sizes(dawn, oracle, n, k, shadow):
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 3)
    for q in Q:
        lab, dist = idx.search(q, k)
        olab, odist = oracle.scan_topk(x, ids, q, k)
        assert len(lab) == min(k, n)
        _assert_same(lab, dist, olab, odist)
    assert idx.stats()["searches"] == 3


def test_scan_1m_batch1_and_batch(dawn, oracle):
    """configs[1]: 1M x 384 f32, batch=1 (and a small batch through the batched entry point)."""
    n = 1_000_000
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 5), synth.planted_queries(1, [0, 12345, n - 1], 9)])
    labels, dist, found = idx.search_batch(Q, 10)
    for b, q in enumerate(Q):
        olab, odist = oracle.scan_topk(x, ids, q, 10, threads=8)
        assert found[b] == 10
        _assert_same(labels[b], dist[b], olab, odist)
    # planted queries: the planted row is the nearest neighbour
    assert labels[5][0] == 1 and labels[6][0] == 12346 and labels[7][0] == n
    lab1, d1 = idx.search(Q[0], 20)  # (an index of this size keeps the packed shadow: its stream)
    olab, odist = oracle.scan_topk(x, ids, Q[0], 20, threads=8)
    _assert_same(lab1, d1, olab, odist)
    idx.set_option("i6_shadow", 0)  # ... and the int8 shadow's
    for q in (Q[0], Q[6]):
        _assert_same(*idx.search(q, 20), *oracle.scan_topk(x, ids, q, 20, threads=8))
    assert idx.stats()["fallbacks"] == 0


def test_exact_fallback_path_agrees(dawn, oracle, shadow):
    """Force the certificate to fail: the always-exact pass must give the same answer."""
    n = 50_000
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, 4)
    want = [oracle.scan_topk(x, ids, q, 20) for q in Q]
    idx.set_option("force_fallback", 1)
    labels, dist, found = idx.search_batch(Q, 20)
    for b in range(4):
        _assert_same(labels[b], dist[b], *want[b])
    assert idx.stats()["fallbacks"] == 4


def test_duplicates_and_ties(dawn, oracle, shadow):
    """Duplicate rows => equal distances => earlier-added row first (KAT from SURVEY §8c).
    200 copies of the best row overflow the 64-entry shortlist band, so the certificate fails and the
    ladder behind it decides — results must still equal the oracle."""
    base = synth.unit_rows(1, 0, 500)
    q = synth.planted_queries(1, [7], 3)[0]
    rows = np.concatenate([base, np.repeat(base[7:8], 200, axis=0), base[:100]])
    ids = np.arange(1000, 1000 + len(rows), dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids, rows)
    lab, dist = idx.search(q, 20)
    olab, odist = oracle.scan_topk(rows, ids, q, 20)
    _assert_same(lab, dist, olab, odist)
    assert lab[0] == 1007 and list(lab[1:5]) == [1500, 1501, 1502, 1503]
    # (with the int8 shadow the bounded exact pass answers a failed certificate — scan_bounded.hip —, without it the exact pass)
    st = idx.stats()
    assert (st["bounded"], st["fallbacks"]) == ((1, 0) if shadow == "i8" else (0, 1))
    # few duplicates: stays on the fast path
    rows2 = np.concatenate([base, base[7:8], base[7:8]])
    ids2 = np.arange(1, len(rows2) + 1, dtype=np.uint64)
    idx2 = dawn.VectorIndex(0)
    idx2.add_batch(ids2, rows2)
    lab, dist = idx2.search(q, 10)
    _assert_same(lab, dist, *oracle.scan_topk(rows2, ids2, q, 10))
    assert list(lab[:3]) == [8, 501, 502]
    assert idx2.stats()["fallbacks"] == 0


def test_known_answers(dawn):
    """query = row => distance ~ 0 and rank 0; antipodal => 2; orthogonal => 1."""
    e = np.zeros((4, 384), dtype=np.float32)
    e[0, 0] = 1.0
    e[1, 1] = 1.0
    e[2, 0] = -1.0
    e[3, 0] = 0.6
    e[3, 1] = 0.8
    idx = dawn.VectorIndex(0)
    idx.add_batch(np.array([10, 11, 12, 13], dtype=np.uint64), e)
    lab, dist = idx.search(e[0], 4)
    assert list(lab) == [10, 13, 11, 12]
    assert dist[0] == 0.0 and dist[2] == 1.0 and dist[3] == 2.0
    assert dist[1] == np.float32(1.0) - np.float32(0.6)


def test_errors_and_empty(dawn):
    idx = dawn.VectorIndex(0)
    q = synth.unit_rows(2, 0, 1)[0]
    lab, dist = idx.search(q, 10)  # empty index: found = 0
    assert len(lab) == 0
    with pytest.raises(dawn.NotNormalizedError):
        idx.search(q * 1.5, 10)
    with pytest.raises(dawn.NotNormalizedError):
        idx.add(1, q * 0.5)
    bad = synth.unit_rows(1, 0, 10)
    bad[4] *= 2.0
    with pytest.raises(dawn.NotNormalizedError):
        idx.add_batch(np.arange(10, dtype=np.uint64), bad)
    assert idx.size() == 0  # nothing added on failure
    nanq = q.copy()
    nanq[3] = np.nan
    with pytest.raises(dawn.NotNormalizedError):
        idx.search(nanq, 10)
    with pytest.raises(dawn.DawnError):
        idx.search(q, 65)
    # boundary of the 0.01 tolerance (vector.rs:185-192)
    assert dawn.is_normalized(q * np.float32(1.009))
    assert not dawn.is_normalized(q * np.float32(1.011))


def test_add_reserve_growth_and_save_load(dawn, oracle, tmp_path):
    idx = dawn.VectorIndex(0)
    rows = synth.unit_rows(5, 0, 3000)
    assert idx.capacity() == 0
    idx.reserve(10)
    assert idx.capacity() == 10 and idx.size() == 0
    for i in range(40):  # single adds across several growth steps (search_provider.rs:280-284)
        if idx.size() == idx.capacity():
            idx.reserve(idx.size() + 16)
        idx.add(100 + i, rows[i])
    idx.add_batch(np.arange(140, 140 + 2960, dtype=np.uint64), rows[40:])
    assert idx.size() == 3000
    got, ids = idx.get_rows(0, 3000)
    assert np.array_equal(got, rows) and np.array_equal(ids, np.arange(100, 3100, dtype=np.uint64))
    q = synth.unit_rows(2, 3, 1)[0]
    want = oracle.scan_topk(rows, ids, q, 20)
    _assert_same(*idx.search(q, 20), *want)
    p = str(tmp_path / "index.dawn")
    idx.save(p)
    idx2 = dawn.VectorIndex(0)
    idx2.load(p)
    assert idx2.size() == 3000
    _assert_same(*idx2.search(q, 20), *want)
    with pytest.raises(dawn.DawnError):
        idx2.load(str(tmp_path / "missing.dawn"))
    assert idx2.size() == 0  # every load failure empties the index (usearch resets before it reads; dawn_hip.h)
    import os
    assert not os.path.exists(p + ".tmp")  # save writes path.tmp, fsyncs and renames


def test_load_is_all_or_nothing(dawn, oracle, tmp_path):
    """A truncated file (an interrupted save of an older build, a full disk) or a row that fails the is_normalized gate must
    leave the index EMPTY, never partially filled: the reference then runs fill_index_from_db() onto it
    (search_provider.rs:115-117) and duplicates would show up in every search."""
    n = 70_000  # more than two staging chunks of the pipelined loader
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    p = str(tmp_path / "index.dawn")
    idx.save(p)
    data = open(p, "rb").read()
    assert len(data) == 24 + n * (8 + 1536)
    q = synth.unit_rows(2, 0, 1)[0]
    want = idx.search(q, 10)
    t = str(tmp_path / "truncated.dawn")
    for cut in (len(data) - 1, 24 + n * 8 + 40_000 * 1536 + 17, 24 + 100, 10):
        open(t, "wb").write(data[:cut])
        idx2 = dawn.VectorIndex(0)
        idx2.fill_synthetic(3, 0, 100, 1)
        with pytest.raises(dawn.DawnError):
            idx2.load(t)
        # EVERY failure empties the index (usearch resets before it reads): the old rows must not survive into the rebuild
        assert idx2.size() == 0 and len(idx2.search(q, 10)[0]) == 0
    idx2 = dawn.VectorIndex(0)
    idx2.fill_synthetic(3, 0, 100, 1)
    with pytest.raises(dawn.DawnError):
        idx2.load(str(tmp_path / "no_such_file.dawn"))
    assert idx2.size() == 0
    idx2.fill_synthetic(3, 0, 100, 1)  # ... and it fills again (what fill_index_from_db does next)
    assert idx2.size() == 100 and len(idx2.search(q, 10)[0]) == 10
    # a bad row far into the file: the rows in front of it were already on the device
    bad = bytearray(data)
    off = 24 + n * 8 + 50_000 * 1536
    bad[off:off + 1536] = (np.frombuffer(data[off:off + 1536], dtype=np.float32) * 3).tobytes()
    open(t, "wb").write(bytes(bad))
    idx3 = dawn.VectorIndex(0)
    with pytest.raises(dawn.NotNormalizedError):
        idx3.load(t)
    assert idx3.size() == 0 and len(idx3.search(q, 10)[0]) == 0
    idx3.load(p)  # ... and the index is still usable
    assert idx3.size() == n
    got = idx3.search(q, 10)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    rows, ids = idx3.get_rows(n - 5, 5)
    assert np.array_equal(rows.view(np.uint32), oracle.unit_rows(1, n - 5, 5).view(np.uint32))
    assert np.array_equal(ids, np.arange(n - 4, n + 1, dtype=np.uint64))


def test_page_entry_file_loader(dawn, oracle, tmp_path):
    """Packed PageEntry records (src/index/warc.rs:35-43): 1568 B, vector at byte 16."""
    n = 777
    rows = synth.unit_rows(6, 0, n)
    rec = np.zeros((n, 1568), dtype=np.uint8)
    rec[:, :8] = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)  # url_pos
    rec[:, 16:16 + 1536] = rows.view(np.uint8).reshape(n, 1536)
    rec[:, 1552:1560] = 33  # url_len garbage
    p = str(tmp_path / "x.warc.emb")
    rec.tofile(p)
    idx = dawn.VectorIndex(0)
    idx.load_page_entries(p, first_id=1)
    assert idx.size() == n
    q = synth.unit_rows(2, 0, 1)[0]
    ids = np.arange(1, n + 1, dtype=np.uint64)
    _assert_same(*idx.search(q, 10), *oracle.scan_topk(rows, ids, q, 10))
    # the reference's own brute-force loop (examples_old/search.rs:49-72) ranks by L2^2 = 2*(1-dot):
    # same first hit
    import ctypes
    ent = np.zeros(10, dtype=np.uintp)
    sc = np.zeros(10, dtype=np.float32)
    m = oracle.lib().orc_scan_examples_old(rec.reshape(-1), n, q, ent, sc)
    assert m == 10
    # ... its ranking is the IP ranking (sum (a-b)^2 = 2 - 2 a.b on unit vectors), entry e = label e + 1, scores 2 x distance
    lab, dist = idx.search(q, 10)
    assert np.array_equal(ent.astype(np.uint64) + 1, lab), (ent, lab)
    assert np.all(np.diff(sc) >= 0) and np.allclose(sc, 2.0 * dist, rtol=0, atol=1e-5)
    # a file with a trailing partial record: entries() = len / size_of::<PageEntry>() (document_embeddings.rs:60-62)
    with open(p, "ab") as f:
        f.write(b"\x01" * 700)
    idx2 = dawn.VectorIndex(0, dtype="bf16")
    idx2.load_page_entries(p, first_id=1)
    assert idx2.size() == n
    # a non-unit vector anywhere in the file: nothing is added
    rec[n // 2, 16:16 + 1536] = (rows[n // 2] * 2).view(np.uint8)
    rec.tofile(p)
    with pytest.raises(dawn.NotNormalizedError):
        idx.load_page_entries(p, first_id=10_000)
    assert idx.size() == n


def test_remote_search_distance_limit(dawn, oracle, shadow):
    """udp_service.rs:196-199: a peer's search reports only the hits with distance < distance_limit (the asker's
    worst_distance(): 0.0 until it holds 20 results, best_results.rs:40 — which prunes everything but negative
    distances, exactly as the reference does)."""
    n = 50_000
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    q = synth.unit_rows(2, 0, 1)[0]
    olab, odist = oracle.scan_topk(x, ids, q, 20)
    for limit in (float(odist[7]), float(np.nextafter(odist[7], np.float32(2))), 5.0, 0.0, float(odist[0])):
        lab, dist = idx.search_limited(q, 20, limit)
        keep = int(np.sum(odist < np.float32(limit)))
        assert len(lab) == keep
        _assert_same(lab, dist, olab[:keep], odist[:keep])


def test_search_provider_mirror(dawn, oracle):
    sp = dawn.SearchProvider(0)
    rows = synth.unit_rows(7, 0, 300)
    for i, r in enumerate(rows):
        sp.insert(dawn.ExtractedPage(url=f"http://x/{i}", title=f"t{i}", text=f"body {i}"), r)
    sp.insert(dawn.ExtractedPage(url="http://x/5"), rows[5])  # duplicate URL: ignored (:254-263)
    assert sp.page_count() == 300 and sp.index.size() == 300
    q = synth.planted_queries(7, [42], 1)[0]
    res = sp.search_embedding(q)
    assert res.pages_searched == 300 and len(res.pages) == 20
    assert res.pages[0].page_id == 43 and res.pages[0].url == "http://x/42"
    olab, odist = oracle.scan_topk(rows, np.arange(1, 301, dtype=np.uint64), q, 20)
    assert [p.page_id for p in res.pages] == [int(v) for v in olab]
    assert np.array_equal(np.array([p.distance for p in res.pages], dtype=np.float32), odist)
    like = sp.search_like(43)
    assert like.pages[0].page_id == 43 and like.pages[0].distance < 0.001  # web.rs:339 "same page"
    with pytest.raises(dawn.NotNormalizedError):
        sp.search_embedding(q * 3)


# ---- batched (matrix-core) path: B > 8 -------------------------------------------------------------
# N <= 8192: one dense pass; 8192 < N <= ~1.5M: dense sample -> threshold -> full append pass;
# larger: dense sample -> appended sample -> full pass (dawn::plan_batched).

@pytest.mark.parametrize("n,B,k", [(100_003, 9, 10), (100_003, 32, 20), (100_003, 33, 10), (50_000, 64, 64),
                                   (50_000, 100, 20), (31, 16, 10), (64, 40, 64), (65, 9, 64), (4097, 256, 20),
                                   (8192, 31, 10), (8193, 200, 20), (8257, 12, 1), (20_001, 256, 10)])
def test_batched_scan_matches_oracle(dawn, oracle, n, B, k, shadow):
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    Q[B // 2] = synth.planted_queries(1, [n // 3], 2)[0]
    Q[B - 1] = x[n - 1]  # exact hit on the last row
    labels, dist, found = idx.search_batch(Q, k)
    for b in range(B):
        olab, odist = oracle.scan_topk(x, ids, Q[b], k, threads=8)
        assert found[b] == min(k, n)
        _assert_same(labels[b][:found[b]], dist[b][:found[b]], olab, odist)
    assert labels[B // 2][0] == n // 3 + 1 and labels[B - 1][0] == n
    # k == shortlist length (64) leaves no margin for the 64-row certificate: those searches are decided by the
    # 1024-row second certificate (every candidate above the threshold rescored exactly), not by the exact pass
    st = idx.stats()
    # k = 64 leaves the 64-row certificate no room: every query takes the 1024-row one.  k = 20: the int8 bound's slack
    # (E ~ 0.009 in the rotated basis, + K2 on a dense-only index) is of the order of the gap between the 20th and the
    # 64th best score, so some queries do; k <= 10: none.  Never an exact pass.
    assert st["fallbacks"] == 0
    if k >= 64 and n > 64:
        assert st["second_chances"] == B
    elif k <= 10 or shadow != "i8":
        assert st["second_chances"] == 0
    else:
        assert st["second_chances"] <= B and st["deepened"] == st["second_chances"]  # settled by a deeper round


def test_batched_1m_batch256(dawn, oracle, shadow):
    """configs[2] scan leg: 1M x 384, batch = 256 — every 5th query checked against the oracle, all 256
    against the streaming path."""
    n, B, k = 1_000_000, 256, 10
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    labels, dist, found = idx.search_batch(Q, k)
    for b in range(0, B, 5):
        olab, odist = oracle.scan_topk(x, ids, Q[b], k, threads=8)
        _assert_same(labels[b], dist[b], olab, odist)
    idx.set_option("mfma_min_batch", 100000)  # force the streaming filter
    l2, d2, f2 = idx.search_batch(Q, k)
    assert np.array_equal(labels, l2) and np.array_equal(dist.view(np.uint32), d2.view(np.uint32))
    assert idx.stats()["fallbacks"] == 0


def test_batched_two_sample_plan_3m(dawn, oracle, shadow):
    """N large enough for the three-pass plan (dense sample, appended sample, full pass); k = 20; B = 300
    exercises the 256-query chunking of the device path."""
    n, B, k = 3_000_000, 300, 20
    idx = _mk_index(dawn, n)
    Q = synth.unit_rows(2, 0, B)
    Q[17] = synth.planted_queries(1, [n - 5], 2)[0]
    labels, dist, found = idx.search_batch(Q, k)
    assert labels[17][0] == n - 4
    idx.set_option("mfma_min_batch", 100000)
    sel = [0, 17, 100, 255, 256, 299]
    l2, d2, f2 = idx.search_batch(Q[sel], k)
    assert np.array_equal(labels[sel], l2) and np.array_equal(dist[sel].view(np.uint32), d2.view(np.uint32))
    assert idx.stats()["fallbacks"] == 0


def test_int8_filter_mfma_shapes_agree(dawn, oracle):
    """The int8 batched filter runs on v_mfma_i32_16x16x64_i8 (default) or, option "mfma_sched" = 32, on
    v_mfma_i32_32x32x32_i8 — same images, same thresholds: identical results (the filter only has to be an upper bound, the
    rescoring is exact), both equal to the oracle; ragged sizes, 1..256 queries, with and without the sampled thresholds."""
    for n, B, k in ((8200, 40, 10), (100_003, 1 + 32, 20), (300_001, 256, 10), (70_000, 17, 64)):
        idx = _mk_index(dawn, n)
        x = oracle.unit_rows(1, 0, n)
        ids = np.arange(1, n + 1, dtype=np.uint64)
        Q = synth.unit_rows(2, 0, B)
        Q[B // 2] = x[n - 1]
        res = {}
        for sched in (32, 4):
            idx.set_option("mfma_sched", sched)
            res[sched] = idx.search_batch(Q, k)
        assert all(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
                   for a, b in zip(res[32], res[4]))
        for b in (0, B // 2, B - 1):
            olab, odist = oracle.scan_topk(x, ids, Q[b], k, threads=8)
            _assert_same(res[4][0][b][:k], res[4][1][b][:k], olab, odist)
        assert idx.stats()["fallbacks"] == 0


def test_batched_filter_error_within_bound(dawn):
    """The certificate assumes |f16 filter score - exact dot| <= FILTER_EPS_F16 = 1.25e-3 (kernels.hpp); measure it
    against float64 on random unit rows, on planted near-duplicates (large scores) and on sparse rows with tiny
    components (f16 subnormal territory before the 2^8 scaling)."""
    rng = np.random.default_rng(5)
    base = synth.unit_rows(1, 0, 6000)
    sparse = np.zeros((1000, 384), dtype=np.float32)
    for i in range(1000):
        sparse[i, rng.integers(0, 384, 3)] = rng.standard_normal(3).astype(np.float32)
        sparse[i] += (1e-6 * rng.standard_normal(384)).astype(np.float32)
        sparse[i] = dawn.normalize(sparse[i])
    rows = np.concatenate([base, sparse])
    idx = dawn.VectorIndex(0)
    idx.set_option("i8_shadow", 0)
    idx.add_batch(np.arange(1, len(rows) + 1, dtype=np.uint64), rows)
    Q = np.concatenate([synth.unit_rows(2, 0, 40), synth.planted_queries(1, [5, 77, 4000], 3), sparse[:5]])
    f = idx.debug_filter_scores(Q)
    assert f.shape == (len(Q), len(rows))
    exact = Q.astype(np.float64) @ rows.astype(np.float64).T
    err = np.abs(f.astype(np.float64) - exact).max()
    assert err < 1.25e-3 / 2, err
    # typical error is far below the worst-case bound (random roundings cancel)
    assert np.abs(f.astype(np.float64) - exact).mean() < 5e-5


def test_int8_batched_scores_are_upper_bounds(dawn):
    """The int8 matrix-core filter (scan_i8.hip) promises filter score >= real dot product for EVERY row (the
    certificates need nothing else): measured against float64 on random rows, on the rows that stretch a per-sub-tile
    quantiser (one-hot, sparse, near-duplicates: _adversarial_rows) and on queries of the same kinds.  The slack of the
    bound stays small on ordinary rows (it decides how often the 64-row certificate holds)."""
    rows, base, extra = _adversarial_rows(7000)
    idx = dawn.VectorIndex(0)
    idx.add_batch(np.arange(1, len(rows) + 1, dtype=np.uint64), rows)
    onehot = np.zeros(384, np.float32); onehot[5] = 1.0
    Q = np.concatenate([synth.unit_rows(2, 0, 40), synth.planted_queries(1, [9, 77, 4000], 3), extra[:12], onehot[None]])
    f = idx.debug_filter_scores(Q)
    assert f.shape == (len(Q), len(rows))
    exact = Q.astype(np.float64) @ rows.astype(np.float64).T
    slack = f.astype(np.float64) - exact
    assert slack.min() > -4e-6, slack.min()  # (-: the f32 rounding of the rotation the shadow lives in, scan_i8.hip)
    # E + K2 ~ 0.017 in the rotated basis — for EVERY kind of row and query: one-hot / sparse rows no longer stretch the
    # quantiser of their sub-tile (before the rotation: up to 0.25)
    assert np.median(slack[:40]) < 0.02 and slack.max() < 0.03, (np.median(slack[:40]), slack.max())


@pytest.mark.parametrize("shadow", ["f16", "i8"])
@pytest.mark.parametrize("n", [127, 5000, 300_001])
def test_stream_filter_lists_hold_the_top64(dawn, n, shadow):
    """The batch-1 streaming filter (MFMA straight from the fragment-ordered shadow) hands merge_rescore one
    descending 64-entry list per workgroup, no row listed twice.
    f16 shadow: every listed score is within FILTER_EPS_F16 of the exact dot of ITS row, and the union of the lists
    holds every row whose exact score clears the 64th best by more than twice that bound.
    int8 shadow: every listed score is an UPPER BOUND of its row's exact dot (scan_i8.hip), at most 0.02 above it on
    this data, and every row whose exact score exceeds T = the largest 64th entry of any list is listed (what the
    certificates rely on: an unlisted row's bound is <= its workgroup's 64th entry)."""
    idx = _mk_index(dawn, n)
    idx.set_option("i8_shadow", int(shadow == "i8"))
    x = synth.unit_rows(1, 0, n)
    for q in list(synth.unit_rows(2, 0, 2)) + [synth.planted_queries(1, [n // 2], 4)[0]]:
        sc, rows = idx.debug_stream_lists(q)
        valid = rows != 0xFFFFFFFF
        assert np.all(np.isneginf(sc[~valid]))
        got = rows[valid].astype(np.int64)
        assert got.max() < n and len(np.unique(got)) == len(got)
        exact = x.astype(np.float64) @ q.astype(np.float64)
        diff = sc[valid].astype(np.float64) - exact[got]
        if shadow == "f16":
            assert np.abs(diff).max() < 1.25e-3 / 2
        else:
            assert diff.min() > -4e-6 and diff.max() < 0.02, (diff.min(), diff.max())
        for b in range(len(sc)):  # descending inside a list, fillers last
            nv = int(valid[b].sum())
            assert np.all(valid[b][:nv]) and np.all(np.diff(sc[b][:nv]) <= 0)
        order = np.argsort(-exact, kind="stable")
        need = order[: min(64, n)]
        if n > 64:
            if shadow == "f16":
                need = need[exact[need] > exact[order[63]] + 2 * 1.25e-3]
            else:
                T = sc[:, 63].max()
                need = np.nonzero(exact > T + 4e-6)[0]
        assert set(need.tolist()) <= set(got.tolist())


def _adversarial_rows(n_base):
    """Synthetic rows + the rows that stretch a per-sub-tile int8 quantiser: one-hot and two-hot rows (scale 1/127 for
    their whole sub-tile), sparse rows, near-duplicates of one row, a row and its negation."""
    rng = np.random.default_rng(11)
    base = synth.unit_rows(1, 0, n_base)
    extra = []
    for j in (0, 5, 383):
        e = np.zeros(384, np.float32); e[j] = 1.0; extra.append(e)
        e = np.zeros(384, np.float32); e[j] = -1.0; extra.append(e)
    e = np.zeros(384, np.float32); e[3] = 0.6; e[200] = 0.8; extra.append(e)
    for nz in (2, 7, 40):
        e = np.zeros(384, np.float32)
        e[rng.choice(384, nz, replace=False)] = rng.standard_normal(nz)
        extra.append((e / np.linalg.norm(e)).astype(np.float32))
    for i in range(40):  # near-duplicates of row 9
        v = base[9] + rng.standard_normal(384).astype(np.float32) * (1e-4 * (i + 1))
        extra.append((v / np.linalg.norm(v)).astype(np.float32))
    extra.append(-base[9])
    extra = np.stack(extra)
    pos = rng.permutation(n_base + len(extra))  # special rows scattered among the sub-tiles
    rows = np.concatenate([base, extra])[pos]
    return np.ascontiguousarray(rows), base, extra


@pytest.mark.parametrize("n_base", [3000, 200_000])
def test_int8_shadow_bounds_and_results_on_adversarial_rows(dawn, oracle, n_base):
    """int8 shadow on rows that hurt a quantiser (one-hot, sparse, near-duplicate): the listed filter scores stay upper
    bounds of the exact dots, and the search results equal the oracle's and the f16-shadow path's, batch 1..3."""
    rows, base, extra = _adversarial_rows(n_base)
    ids = np.arange(1, len(rows) + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids, rows)
    onehot = np.zeros(384, np.float32); onehot[5] = 1.0
    Q = np.stack([synth.unit_rows(2, 0, 1)[0], onehot, extra[7], base[9], synth.planted_queries(1, [9], 5)[0], -base[9]])
    for q in Q:
        sc, lr = idx.debug_stream_lists(q)
        valid = lr != 0xFFFFFFFF
        got = lr[valid].astype(np.int64)
        exact = rows[got].astype(np.float64) @ q.astype(np.float64)
        assert (sc[valid].astype(np.float64) - exact).min() > -4e-6
    for k in (10, 20):
        for B in (1, 2, 3):
            for j in range(0, len(Q), B):
                qb = Q[j:j + B]
                idx.set_option("i8_shadow", 1)
                l1, d1, f1 = idx.search_batch(qb, k)
                idx.set_option("i8_shadow", 0)
                l0, d0, f0 = idx.search_batch(qb, k)
                assert np.array_equal(l0, l1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32))
                for b in range(len(qb)):
                    _assert_same(l1[b], d1[b], *oracle.scan_topk(rows, ids, qb[b], k))


def test_int8_shadow_tracks_adds_and_growth(dawn, oracle):
    """Rows added in ragged batches (the last sub-tile is re-quantised with its new rows, growth re-quantises all):
    every search in between equals the oracle."""
    x = synth.unit_rows(1, 0, 5000)
    ids = np.arange(1, 5001, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    q = synth.planted_queries(1, [3], 5)[0]
    done = 0
    for step in (1, 30, 1, 33, 64, 1000, 7, 2864, 1000):
        idx.add_batch(ids[done:done + step], x[done:done + step])
        done += step
        for qq in (q, synth.unit_rows(2, done, 1)[0]):
            lab, dist = idx.search(qq, 10)
            _assert_same(lab, dist, *oracle.scan_topk(x[:done], ids[:done], qq, 10))
    assert done == 5000 and idx.stats()["fallbacks"] == 0


@pytest.mark.parametrize("dtype", ["f32", "f32-f16shadow", "bf16"])
@pytest.mark.parametrize("n,B", [(4097, 2), (100_003, 3), (100_003, 5), (300_001, 8), (20_001, 13)])
def test_stream_filter_takes_up_to_8_queries_per_pass(dawn, oracle, n, B, dtype):
    """The streaming filter serves single queries by default; forced (mfma_min_batch) it takes any batch, 8 queries per
    pass (QB = 1 / 4 / 8 variants; int8 shadow, f16 shadow or bf16 index): same results as the oracle and as the
    default path."""
    idx = dawn.VectorIndex(0, dtype=dtype.split("-")[0])
    if dtype == "f32-f16shadow":
        idx.set_option("i8_shadow", 0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    if dtype == "bf16":
        x = synth.round_bf16(x)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    Q[B - 1] = synth.planted_queries(1, [n // 2], 6)[0]
    l0, d0, f0 = idx.search_batch(Q, 10)
    idx.set_option("mfma_min_batch", 100000)
    l1, d1, f1 = idx.search_batch(Q, 10)
    assert np.array_equal(l0, l1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(f0, f1)
    for b in range(B):
        _assert_same(l1[b], d1[b], *oracle.scan_topk(x, ids, Q[b], 10))
    assert l1[B - 1][0] == n // 2 + 1
    assert idx.stats()["fallbacks"] == 0


def test_small_batches_choose_their_path_by_index_size(dawn, oracle):
    """"mfma_min_batch" = 0 (the default): 2-4 queries on an index of (B - 1) x 1.25 M rows and more take ONE stream of the int8 shadow,
    smaller indexes and larger batches the matrix-core pass (dawn_index.cpp: index_search_on_device).  Same answers either way, and
    the oracle's; which path ran shows in the profile: the streaming filter is one launch per search, the pass runs behind its
    sampling passes."""
    n = 1_400_000
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    Q = synth.unit_rows(2, 0, 5)
    Q[1] = synth.planted_queries(1, [n // 3], 6)[0]
    for B in (2, 3, 5):
        want = oracle.scan_topk_synth(1, 0, n, 1, Q[:B], 10)
        got = {}
        for mmb in (0, 2, 100000):
            idx.set_option("mfma_min_batch", mmb)
            got[mmb] = idx.search_batch(Q[:B], 10)
            for b in range(B):
                _assert_same(got[mmb][0][b], got[mmb][1][b], want[0][b], want[1][b])
        assert got[0][0][1][0] == n // 3 + 1
    assert idx.stats()["fallbacks"] == 0
    with pytest.raises(dawn.DawnError):
        idx.set_option("mfma_min_batch", -1)


def test_batched_duplicates_second_certificate_then_exact_pass(dawn, oracle, shadow):
    base = synth.unit_rows(1, 0, 3000)
    rows = np.concatenate([base, np.repeat(base[11:12], 300, axis=0), base[:50]])
    ids = np.arange(1, len(rows) + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids, rows)
    Q = synth.unit_rows(2, 0, 16)
    Q[3] = synth.planted_queries(1, [11], 8)[0]
    labels, dist, found = idx.search_batch(Q, 20)
    for b in range(16):
        _assert_same(labels[b], dist[b], *oracle.scan_topk(rows, ids, Q[b], 20))
    assert labels[3][0] == 12 and list(labels[3][1:4]) == [3001, 3002, 3003]
    # 301 identical rows at the top: the 64-row certificate cannot hold, the 1024-row one does — no exact pass
    st = idx.stats()
    assert st["fallbacks"] == 0 and (st["second_chances"] == 1 if shadow != "i8" else 1 <= st["second_chances"] <= 8)
    # 1500 identical rows: more than the round-3 second certificate looked at (1024) — the second chance now rescores EVERY
    # candidate of the pass (up to 8192): still no pass over the index
    rows3 = np.concatenate([base, np.repeat(base[11:12], 1500, axis=0)])
    ids3 = np.arange(1, len(rows3) + 1, dtype=np.uint64)
    idx3 = dawn.VectorIndex(0)
    idx3.add_batch(ids3, rows3)
    l3, d3, f3 = idx3.search_batch(Q, 20)
    for b in (0, 3, 15):
        _assert_same(l3[b], d3[b], *oracle.scan_topk(rows3, ids3, Q[b], 20))
    st3 = idx3.stats()
    assert (st3["bounded"], st3["fallbacks"]) == (0, 0) and st3["second_chances"] >= 1
    # 9000 identical rows overflow the candidate buffer itself: the ladder decides — the bounded exact pass where the int8 shadow
    # is kept, the exact pass over all rows otherwise —, same answer
    rows4 = np.concatenate([base, np.repeat(base[11:12], 9000, axis=0)])
    ids4 = np.arange(1, len(rows4) + 1, dtype=np.uint64)
    idx4 = dawn.VectorIndex(0)
    idx4.add_batch(ids4, rows4)
    l4, d4, f4 = idx4.search_batch(Q, 20)
    for b in (0, 3, 15):
        _assert_same(l4[b], d4[b], *oracle.scan_topk(rows4, ids4, Q[b], 20))
    st4 = idx4.stats()
    assert (st4["bounded"], st4["fallbacks"]) == ((1, 0) if shadow == "i8" else (0, 1))


def test_batched_clustered_index_overflow_falls_back(dawn, oracle, shadow):
    """An index the strided sample misrepresents: 256 tiles of 64 rows, even tiles random, odd tiles near-copies
    of one row.  The sample (every 2nd tile) sees only random rows, so a query next to the copied row collects
    > 8192 candidates -> buffer overflow -> exact pass.  Answers must still equal the oracle."""
    base = synth.unit_rows(1, 0, 8192)
    rng = np.random.default_rng(9)
    centre = base[123]
    near = centre[None, :] + (0.02 / np.sqrt(384)) * rng.standard_normal((8192, 384)).astype(np.float32)
    near = np.stack([dawn.normalize(v) for v in near.astype(np.float32)])
    rows = np.empty((16384, 384), dtype=np.float32)
    rows.reshape(128, 2, 64, 384)[:, 0] = base.reshape(128, 64, 384)
    rows.reshape(128, 2, 64, 384)[:, 1] = near.reshape(128, 64, 384)
    ids = np.arange(1, len(rows) + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids, rows)
    Q = synth.unit_rows(2, 0, 12)
    Q[4] = centre
    labels, dist, found = idx.search_batch(Q, 10)
    for b in range(12):
        _assert_same(labels[b], dist[b], *oracle.scan_topk(rows, ids, Q[b], 10, threads=8))
    assert labels[4][0] == 2 * 64 * (123 // 64) + 123 % 64 + 1  # where base[123] landed
    st = idx.stats()
    assert st["fallbacks"] + st["bounded"] >= 1 and (st["fallbacks"] == 0 if shadow == "i8" else st["bounded"] == 0)


def test_batched_correlated_queries_burst(dawn, oracle, shadow):
    """256 near-identical queries (a realistic batch: one topic) hit the SAME rows, so a qualifying row appends 256
    candidates at once.  The workgroup's candidate stage must absorb or bypass such bursts without dropping
    anything or sending the batch to the exact pass."""
    n, B, k = 200_000, 256, 10
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    base = synth.unit_rows(2, 0, 1)[0]
    rng = np.random.default_rng(3)
    Q = np.stack([dawn.normalize(base + (0.01 / np.sqrt(384)) * rng.standard_normal(384).astype(np.float32))
                  for _ in range(B)])
    labels, dist, found = idx.search_batch(Q, k)
    for b in range(0, B, 17):
        _assert_same(labels[b], dist[b], *oracle.scan_topk(x, ids, Q[b], k, threads=8))
    assert idx.stats()["fallbacks"] == 0
    idx.set_option("mfma_sched", 0)  # the lockstep kernel (on-the-fly conversion path) as well
    idx.set_option("f16_shadow", 0)
    l2, d2, f2 = idx.search_batch(Q, k)
    assert np.array_equal(labels, l2) and np.array_equal(dist.view(np.uint32), d2.view(np.uint32))
    assert idx.stats()["fallbacks"] == 0
    idx.set_option("f16_shadow", 1)
    idx.set_option("mfma_sched", 5)  # ... and the pipelined kernel (per-wave stage: a burst must flush and go on)
    try:
        l3, d3, f3 = idx.search_batch(Q, k)
        assert np.array_equal(labels, l3) and np.array_equal(dist.view(np.uint32), d3.view(np.uint32))
        assert idx.stats()["fallbacks"] == 0
    finally:
        idx.set_option("mfma_sched", 4)  # process-wide setting: back to the default


@pytest.mark.parametrize("n,B,k", [(64, 16, 10), (4097, 256, 10), (8193, 9, 20), (20_001, 200, 10), (100_003, 33, 10),
                                   (1_000_000, 256, 10), (3_000_000, 130, 20)])
def test_batched_pipelined_kernel_matches(dawn, oracle, n, B, k):
    """The software-pipelined 4-wave kernel (default for long passes only) forced onto every pass (mfma_sched 5):
    same results as the 8-wave kernel for every query, and as the oracle for a sample; sizes cover one tile, odd
    tile counts, 1 / 2 live query groups per wave, idle waves, both sampling plans."""
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    Q[B - 1] = x[n - 1]
    idx.set_option("mfma_sched", 1)
    try:
        l1, d1, f1 = idx.search_batch(Q, k)
        idx.set_option("mfma_sched", 5)
        l5, d5, f5 = idx.search_batch(Q, k)
    finally:
        idx.set_option("mfma_sched", 4)
    assert np.array_equal(f1, f5) and np.array_equal(l1, l5) and np.array_equal(d1.view(np.uint32), d5.view(np.uint32))
    for b in list(range(0, B, max(1, B // 6))) + [B - 1]:
        _assert_same(l5[b][:f5[b]], d5[b][:f5[b]], *oracle.scan_topk(x, ids, Q[b], k, threads=8))
    assert l5[B - 1][0] == n
    assert idx.stats()["fallbacks"] == 0


def test_batched_shadow_tracks_adds_and_growth(dawn, oracle, shadow):
    """The f16 shadow copy is built at the first batched search and must follow later adds / reallocation."""
    rows = oracle.unit_rows(1, 0, 30_000)
    ids = np.arange(1, 30_001, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids[:9_000], rows[:9_000])
    Q = synth.unit_rows(2, 0, 40)
    Q[7] = rows[29_999]
    l, d, f = idx.search_batch(Q, 10)                      # builds the shadow for 9000 rows
    for b in (0, 7, 39):
        _assert_same(l[b], d[b], *oracle.scan_topk(rows[:9_000], ids[:9_000], Q[b], 10))
    idx.add_batch(ids[9_000:], rows[9_000:])               # grows the index: shadow must be extended + reallocated
    l, d, f = idx.search_batch(Q, 10)
    for b in (0, 7, 39):
        _assert_same(l[b], d[b], *oracle.scan_topk(rows, ids, Q[b], 10, threads=4))
    assert l[7][0] == 30_000 and idx.stats()["fallbacks"] == 0


def test_memory_accounting(dawn):
    """dawn_index_memory: 1536 B per reserved f32 row + 384 B per row (+ 8 B per 32 rows) of int8 shadow, which every
    mutation keeps current (a search allocates nothing), and + 768 B per row once the f16 shadow is asked for."""
    n = 100_000
    idx = dawn.VectorIndex(0)
    idx.reserve(n)
    m0 = idx.memory()
    assert m0["rows"] >= n * 1536 and m0["rows"] < 1.6 * n * 1536 and m0["shadows"] == 0
    idx.fill_synthetic(1, 0, n, 1)
    q = synth.unit_rows(2, 0, 1)[0]
    m1 = idx.memory()
    idx.search(q, 10)
    assert idx.memory() == m1
    per_row = m1["shadows"] / (m1["rows"] / 1536)
    assert 384 <= per_row < 386, per_row
    idx.set_option("i8_shadow", 0)
    m2 = idx.memory()
    idx.search(q, 10)
    assert idx.memory() == m2
    assert m2["shadows"] - m1["shadows"] == (m1["rows"] // 1536) * 768
    assert m2["other"] > n * 8


@pytest.mark.parametrize("n", [127, 128, 129, 255, 257, 8191, 8192, 8193, 8320, 8321, 16385, 40_001])
@pytest.mark.parametrize("B", [4, 130, 256])
def test_int8_batched_tile_edges(dawn, oracle, n, B):
    """Index sizes around the 128-row tiles and 32-row sub-tiles of the int8 matrix-core pass, around the dense-only
    limit (8192 rows) and the smallest sampled plans; query counts that use one wave, five waves / both groups, all of
    them.  Bit-identical to the oracle; the rows past the end of the last tile never show up."""
    idx = _mk_index(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows(2, 0, B)
    Q[B - 1] = x[n - 1]          # the very last row
    Q[0] = synth.planted_queries(1, [n - (n % 32 or 32)], 3)[0]  # first row of the last sub-tile
    labels, dist, found = idx.search_batch(Q, 10)
    for b in list(range(0, B, max(1, B // 12))) + [B - 1]:
        olab, odist = oracle.scan_topk(x, ids, Q[b], 10, threads=8)
        assert found[b] == min(10, n)
        _assert_same(labels[b][:found[b]], dist[b][:found[b]], olab, odist)
    assert labels[B - 1][0] == n and labels.max() <= n
    assert idx.stats()["fallbacks"] == 0


def test_heavy_tailed_golden_fixture_all_paths(dawn, oracle):
    """The committed scan fixture on bell-shaped rows with four heavy dimensions (tests/golden/scan_normal_seed4.npz, by
    the numpy restatement): streaming filter, matrix-core pass, both shadows, a bf16-free f32 index — bit-identical, and
    without an exact pass: the rotation of the int8 shadow takes the heavy dimensions out of the quantiser's way
    (scan_i8.hip; without it these rows sent most queries to the exact pass)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "scan_normal_seed4.npz"))
    n = int(g["n_rows"])
    X = synth.unit_rows_normal(int(g["index_seed"]), 0, n, heavy_dims=tuple(int(d) for d in g["heavy_dims"]))
    ids = np.arange(1, n + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids, X)
    Q = g["queries"]
    for i8 in (1, 0):
        idx.set_option("i8_shadow", i8)
        lab, dist, found = idx.search_batch(Q, 20)
        assert np.array_equal(lab, g["labels"]) and np.array_equal(dist.view(np.uint32), g["distances"].view(np.uint32))
        for b in (0, 3):
            l, d = idx.search(Q[b], 20)
            assert np.array_equal(l, g["labels"][b]) and np.array_equal(d.view(np.uint32), g["distances"][b].view(np.uint32))
    assert idx.stats()["fallbacks"] == 0
    # the bound's slack on these rows equals the slack on any others (what the rotation is for)
    idx.set_option("i8_shadow", 1)
    f = idx.debug_filter_scores(Q)
    slack = f.astype(np.float64) - Q.astype(np.float64) @ X[:f.shape[1]].astype(np.float64).T
    assert slack.min() > -4e-6 and slack.max() < 0.03, (slack.min(), slack.max())


def test_single_row_adds_are_staged_and_flushed_in_order(dawn, oracle, tmp_path):
    """dawn_index_add stages rows on the host (1024 at a time) and flushes them in front of whatever looks at the rows next:
    the reference's one-row-per-call loops (search_provider.rs:127-153,280-284) see exactly the old semantics — size() counts
    every added row at once, a search right after an add finds it, order is insertion order, a bad row is refused on the spot
    and nothing else is lost, load() replaces staged rows too."""
    rows = synth.unit_rows(8, 0, 2600)
    ids = np.arange(7000, 7000 + 2600, dtype=np.uint64)
    for devices in (None, [0, 0, 0]):
        idx = dawn.VectorIndex(0) if devices is None else dawn.VectorIndex(devices=devices)
        if devices is not None:
            idx.set_option("shard_chunk", 64)
        for i in range(2600):
            if idx.size() == idx.capacity():
                idx.reserve(idx.size() + 1024)
            idx.add(int(ids[i]), rows[i])
            assert idx.size() == i + 1
            if i in (0, 5, 1023, 1024, 1500, 2599):  # a search right behind an add sees the row
                lab, dist = idx.search(rows[i], 3)
                assert lab[0] == ids[i] and dist[0] < 1e-6
            if i == 700:
                with pytest.raises(dawn.NotNormalizedError):
                    idx.add(1, rows[0] * np.float32(1.5))
                assert idx.size() == 701
        q = synth.unit_rows(2, 5, 1)[0]
        want = oracle.scan_topk(rows, ids, q, 20)
        _assert_same(*idx.search(q, 20), *want)
        got, gids = idx.get_rows(0, 2600)
        assert np.array_equal(got, rows) and np.array_equal(gids, ids)
        # staged rows (not flushed yet) are part of a save, and are replaced by a load
        extra = synth.unit_rows(9, 0, 10)
        for j in range(10):
            idx.add(9000 + j, extra[j])
        p = str(tmp_path / f"staged_{0 if devices is None else 1}.dawn")
        idx.save(p)
        other = dawn.VectorIndex(0)
        other.load(p)
        assert other.size() == 2610 and other.search(extra[9], 1)[0][0] == 9009
        for j in range(5):
            other.add(1 + j, rows[j])  # staged ...
        assert other.size() == 2615
        other.load(p)  # ... and dropped: load replaces the contents
        assert other.size() == 2610
        # batches and single adds interleave in insertion order (ties -> earlier-added row)
        dup = dawn.VectorIndex(0)
        dup.add(5, rows[3])
        dup.add_batch(np.array([4, 3], dtype=np.uint64), np.stack([rows[3], rows[3]]))
        dup.add(2, rows[3])
        assert dup.search(rows[3], 4)[0].tolist() == [5, 4, 3, 2]


def test_out_of_memory_order_of_the_filter_sources(dawn, oracle):
    """When HBM runs out the filters fall back int8 shadow -> f16 shadow -> the f32 rows themselves (a 100 M-row f32 index with
    both shadows needs 230 GB).  "debug_fail_alloc" makes the allocations fail as a full card would: every stage of the order
    must still answer bit-identically to the oracle, for single queries and for batches, and report what it holds."""
    n = 150_000
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 11), synth.planted_queries(1, [31_337], 4)])
    want = [oracle.scan_topk(x, ids, q, 20, threads=4) for q in Q]
    seen = []
    for mask in (0, 1, 3, 2, 0):  # normal; int8 fails -> f16; both fail -> rows; f16 fails alone (int8 carries everything); normal
        idx.set_option("debug_fail_alloc", mask)
        seen.append(idx.memory()["shadows"])
        lab, dist, found = idx.search_batch(Q, 20)
        for b in range(len(Q)):
            assert np.array_equal(lab[b], want[b][0]) and np.array_equal(dist[b].view(np.uint32), want[b][1].view(np.uint32)), (mask, b)
        for b in (0, 11):
            _assert_same(*idx.search(Q[b], 20), *want[b])
        # rows added while a shadow is missing keep every path exact
        extra = synth.unit_rows(9, mask * 10, 3)
        for j in range(3):
            idx.add(10_000_000 + mask * 10 + j, extra[j])
        l1, d1 = idx.search(extra[1], 1)
        assert l1[0] == 10_000_000 + mask * 10 + 1
        x = np.concatenate([x, extra])
        ids = np.concatenate([ids, np.arange(10_000_000 + mask * 10, 10_000_000 + mask * 10 + 3, dtype=np.uint64)])
        want = [oracle.scan_topk(x, ids, q, 20, threads=4) for q in Q]
    i8_bytes, f16_bytes = n * 384, n * 768
    assert seen[0] >= i8_bytes and seen[0] < f16_bytes          # int8 shadow only
    assert seen[1] >= f16_bytes and seen[1] < 2 * f16_bytes          # f16 shadow only
    assert seen[2] == 0                                          # nothing: the filters stream the f32 rows
    assert seen[3] >= i8_bytes and seen[3] < f16_bytes and seen[4] >= i8_bytes
    assert idx.stats()["fallbacks"] == 0


@pytest.mark.parametrize("blocks", [256, 40, 7])
def test_f32_stream_dynamic_tail_covers_every_row(dawn, oracle, blocks):
    """The f32-row stream (shadows switched off for single queries) hands the last eighth of a long launch out on demand in
    blocks of 16 iterations: whatever the grid, every row is scanned exactly once — queries planted all over the index come back
    first — and the answers equal the oracle's and the static assignment's (option "stream_dynamic_tail" = 0)."""
    n = 1_200_003
    idx = _mk_index(dawn, n)
    idx.set_option("f16_shadow_b1", 0)
    idx.set_option("scan_blocks", blocks)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    planted = np.array([0, 5, 600_000, 1_049_999, 1_050_001, 1_100_017, 1_150_000, n - 40_000, n - 7, n - 1])
    Q = synth.planted_queries(1, planted, 5)
    for r, q in zip(planted, Q):
        lab, dist = idx.search(q, 10)
        assert lab[0] == r + 1, (r, lab)
        _assert_same(lab, dist, *oracle.scan_topk(x, ids, q, 10, threads=8))
    q = synth.unit_rows(2, 0, 1)[0]
    want = oracle.scan_topk(x, ids, q, 20, threads=8)
    for mode in (1, 0, 1):
        idx.set_option("stream_dynamic_tail", mode)
        for _ in range(2):  # (twice: the counters are back at zero after every search)
            _assert_same(*idx.search(q, 20), *want)
    assert idx.stats()["fallbacks"] == 0


@pytest.mark.parametrize("blocks,n", [(8, 300_000), (5, 200_003), (16, 600_000)])
def test_int8_pass_long_tile_sequences_on_small_grids(dawn, oracle, blocks, n):
    """The int8 matrix-core pass on a small grid (option "mfma_blocks"): every workgroup walks a long, strided sequence of tiles
    (>= 256 of them) — planted rows in the first tile, in the middle, in the last, ragged tile come back first —, results equal
    to the oracle's, batch after batch."""
    idx = _mk_index(dawn, n)
    idx.set_option("mfma_blocks", blocks)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    B = 70
    Q = synth.unit_rows(2, 0, B)
    planted = np.array([0, 127, 128, n // 2, (n * 7) // 8 - 3, (n * 7) // 8 + 300, (n * 15) // 16, n - 129, n - 2, n - 1])
    Q[:len(planted)] = synth.planted_queries(1, planted, 5)
    ref = None
    for _ in range(3):
        labels, dist, found = idx.search_batch(Q, 20)
        assert np.all(found == 20) and np.array_equal(labels[:len(planted), 0], planted + 1)
        if ref is None:
            ref = (labels, dist)
            for b in (0, 3, 5, 9, 20, 69):
                _assert_same(labels[b], dist[b], *oracle.scan_topk(x, ids, Q[b], 20, threads=8))
        assert np.array_equal(labels, ref[0]) and np.array_equal(dist.view(np.uint32), ref[1].view(np.uint32))
    assert idx.stats()["fallbacks"] == 0
