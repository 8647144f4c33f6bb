"""GPU parity tests of the packed (5- or 6-bit) filter shadow and its single-query stream (dawnsearch_amd/csrc/scan_i6.hip).

The default single-query search of an index of >= 768 Ki rows streams a 5-bit copy of the rows (240 B per row; 6 bits = 288 B
selectable), rescoring every workgroup's 64-row shortlists exactly in the kernel's epilogue and merging the exact lists under
one certificate.  Here the path is forced on small indexes (option "i6_min_rows" = 0), in both widths (fixture `bits`: env
DAWN_I6_BITS, read when an index is created), and held against the CPU oracle (oracle/dawn_oracle.c, a restatement
of src/search/vector.rs:128-134 + exact top-k): BIT-EXACT distances, identical label order, as for every other path.  The
full-size checks (100 M rows against the oracle's own scan) are in test_full_size_gpu.py.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from dawnsearch_amd import synth  # noqa: E402

from test_scan_gpu import _adversarial_rows, _assert_same  # noqa: E402


@pytest.fixture(params=[5, 6], autouse=True)
def bits(request, monkeypatch):
    monkeypatch.setenv("DAWN_I6_BITS", str(request.param))
    return request.param


def _mk(dawn, n, dtype="f32", seed=1):
    idx = dawn.VectorIndex(0, dtype=dtype)
    idx.set_option("i6_min_rows", 0)
    idx.fill_synthetic(seed, 0, n, 1)
    return idx


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 127, 1000, 4097, 100_003, 300_001])
@pytest.mark.parametrize("k", [1, 10, 20, 64])
def test_i6_stream_matches_oracle_sizes(dawn, oracle, n, k):
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [n // 2], 4)])
    for q in Q:
        lab, dist = idx.search(q, k)
        assert len(lab) == min(k, n)
        _assert_same(lab, dist, *oracle.scan_topk(x, ids, q, k))
    assert idx.stats()["searches"] == 4
    # (k = 64 of a small index whose rows all fall to one or two workgroups: the 64th listed bound is the bound of the 65th
    # row too, the certificate fails by construction and the exact pass answers — as for every other stream)
    if k <= 20 and (n <= 64 or n >= 1000):
        assert idx.stats()["fallbacks"] == 0


def test_i6_is_the_default_of_large_indexes_only(dawn, oracle, bits):
    """Below i6_min_rows (768 Ki by default) no packed shadow is built; the option / the size crossing the limit builds it, and
    switching it off gives the memory back.  Same results either way."""
    n = 200_000
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    m0 = idx.memory()["shadows"]
    q = synth.planted_queries(1, [77], 3)[0]
    want = idx.search(q, 10)
    idx.set_option("i6_min_rows", 100_000)
    m1 = idx.memory()["shadows"]
    rb = 288 if bits == 6 else 240
    assert m1 - m0 >= n * rb and m1 - m0 < (n + 4096) * (rb + 2)
    got = idx.search(q, 10)
    _assert_same(got[0], got[1], want[0], want[1])
    sc, _ = idx.debug_stream_lists(q)
    assert len(sc) == 256  # the stream's lists: one per workgroup
    idx.set_option("i6_shadow", 0)
    assert idx.memory()["shadows"] == m0
    got = idx.search(q, 10)
    _assert_same(got[0], got[1], want[0], want[1])
    idx.set_option("i6_shadow", 1)
    assert idx.memory()["shadows"] == m1
    x = oracle.unit_rows(1, 0, n)
    _assert_same(*idx.search(q, 20), *oracle.scan_topk(x, np.arange(1, n + 1, dtype=np.uint64), q, 20))


@pytest.mark.parametrize("n", [127, 5000, 300_001])
def test_i6_lists_are_upper_bounds_and_cover_everything_above_T(dawn, n, bits):
    """What the certificate of merge_exact_kernel relies on: every listed score is an UPPER BOUND of its row's exact dot (within
    the rounding allowance that is part of FILTER_EPS_I8) — the int8 shadow's bound, which the epilogue puts in the place of the
    packed shadow's: slack < 0.02 —, each list is descending, no row is listed twice, and every row whose exact score exceeds the
    certificate bound T (the largest bound any workgroup gives for its unlisted rows: the coarse bound of what its waves dropped,
    the int8 bound of what its merge dropped) IS listed."""
    idx = _mk(dawn, n)
    x = synth.unit_rows(1, 0, n)
    for q in list(synth.unit_rows(2, 0, 2)) + [synth.planted_queries(1, [n // 2], 4)[0]]:
        sc, rows = idx.debug_stream_lists(q)
        valid = rows != 0xFFFFFFFF
        assert np.all(np.isneginf(sc[~valid]))
        got = rows[valid].astype(np.int64)
        assert got.max() < n and len(np.unique(got)) == len(got)
        exact = x.astype(np.float64) @ q.astype(np.float64)
        diff = sc[valid].astype(np.float64) - exact[got]
        assert diff.min() > -4e-6 and diff.max() < 0.02, (diff.min(), diff.max())
        for b in range(len(sc)):
            nv = int(valid[b].sum())
            assert np.all(valid[b][:nv]) and np.all(np.diff(sc[b][:nv]) <= 0)
        T = idx.debug_stream_bound()
        if n > 64:
            assert T >= sc[:, 63].max()
            need = np.nonzero(exact > T + 4e-6)[0]
            assert set(need.tolist()) <= set(got.tolist())
        else:
            assert len(got) == n and np.isneginf(T)


@pytest.mark.parametrize("n_base", [3000, 200_000])
def test_i6_bounds_and_results_on_adversarial_rows(dawn, oracle, n_base):
    """Rows that hurt a per-sub-tile quantiser (one-hot, sparse, near-duplicates, a row and its negation): the bounds hold and
    the results equal the oracle's and the int8 path's."""
    rows, base, extra = _adversarial_rows(n_base)
    ids = np.arange(1, len(rows) + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
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
        for q in Q:
            idx.set_option("i6_shadow", 1)
            l1, d1 = idx.search(q, k)
            idx.set_option("i6_shadow", 0)
            l0, d0 = idx.search(q, k)
            _assert_same(l1, d1, l0, d0)
            _assert_same(l1, d1, *oracle.scan_topk(rows, ids, q, k))


def test_i6_duplicates_fail_the_certificate_and_stay_exact(dawn, oracle):
    """More equal rows at the top than the workgroups list (20 000 copies of the best row: > 64 per workgroup): the bound T of
    the unlisted rows reaches the k-th score, the certificate fails, the bounded exact pass on the int8 shadow answers (no pass
    over all rows) — earlier-added rows first."""
    base = synth.unit_rows(1, 0, 500)
    q = synth.planted_queries(1, [7], 3)[0]
    rows = np.concatenate([base, np.repeat(base[7:8], 20_000, axis=0), base[:100]])
    ids = np.arange(1000, 1000 + len(rows), dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
    idx.add_batch(ids, rows)
    lab, dist = idx.search(q, 20)
    _assert_same(lab, dist, *oracle.scan_topk(rows, ids, q, 20))
    assert lab[0] == 1007 and list(lab[1:5]) == [1500, 1501, 1502, 1503]
    st = idx.stats()
    assert st["bounded"] == 1 and st["fallbacks"] == 0
    # ... and with that rung switched off the exact pass over all rows does, as in round 3
    idx.set_option("bounded_pass", 0)
    _assert_same(*idx.search(q, 20), *oracle.scan_topk(rows, ids, q, 20))
    st = idx.stats()
    assert st["bounded"] == 1 and st["fallbacks"] == 1
    # a few duplicates stay on the fast path
    rows2 = np.concatenate([base, base[7:8], base[7:8]])
    ids2 = np.arange(1, len(rows2) + 1, dtype=np.uint64)
    idx2 = dawn.VectorIndex(0)
    idx2.set_option("i6_min_rows", 0)
    idx2.add_batch(ids2, rows2)
    lab, dist = idx2.search(q, 10)
    _assert_same(lab, dist, *oracle.scan_topk(rows2, ids2, q, 10))
    assert list(lab[:3]) == [8, 501, 502] and idx2.stats()["fallbacks"] == 0


def test_i6_forced_fallback_and_distance_limit(dawn, oracle):
    n = 50_000
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    q = synth.unit_rows(2, 0, 1)[0]
    want = oracle.scan_topk(x, ids, q, 20)
    lab, dist = idx.search_limited(q, 20, float(want[1][9]))
    keep = want[1] < want[1][9]  # udp_service.rs:196-199: strictly closer than the limit
    _assert_same(lab, dist, want[0][keep], want[1][keep])
    idx.set_option("force_fallback", 1)
    _assert_same(*idx.search(q, 20), *want)
    assert idx.stats()["fallbacks"] == 1


def test_i6_shadow_tracks_adds_and_growth(dawn, oracle):
    """Ragged adds (the last sub-tile is re-quantised with its new rows, growth re-quantises everything) and single-row adds
    staged on the host: every search in between equals the oracle."""
    x = synth.unit_rows(1, 0, 5000)
    ids = np.arange(1, 5001, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
    q = synth.planted_queries(1, [3], 5)[0]
    done = 0
    for step in (1, 30, 1, 33, 64, 1000, 7, 2864, 990):
        idx.add_batch(ids[done:done + step], x[done:done + step])
        done += step
        for qq in (q, synth.unit_rows(2, done, 1)[0]):
            _assert_same(*idx.search(qq, 10), *oracle.scan_topk(x[:done], ids[:done], qq, 10))
    for i in range(done, 5000):
        idx.add(int(ids[i]), x[i])
    _assert_same(*idx.search(q, 20), *oracle.scan_topk(x, ids, q, 20))
    assert idx.size() == 5000 and idx.stats()["fallbacks"] == 0


@pytest.mark.parametrize("threads,ring", [(64, 12), (128, 6), (192, 8), (256, 4), (320, 3), (384, 6), (512, 12), (512, 4)])
def test_i6_geometries_agree(dawn, oracle, threads, ring):
    n = 150_001
    idx = _mk(dawn, n)
    idx.set_option("i6_scan_threads", threads)
    idx.set_option("i6_scan_ring", ring)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    for q in np.concatenate([synth.unit_rows(2, 0, 2), synth.planted_queries(1, [n - 1], 4)]):
        _assert_same(*idx.search(q, 20), *oracle.scan_topk(x, ids, q, 20))
    idx.set_option("i6_scan_blocks", 40)
    q = synth.unit_rows(2, 5, 1)[0]
    _assert_same(*idx.search(q, 10), *oracle.scan_topk(x, ids, q, 10))
    assert idx.stats()["fallbacks"] == 0


def test_i6_on_a_bf16_index(dawn, oracle, bits):
    """A bf16 index keeps a packed shadow of its (bf16-rounded) rows; the exact side scores the rows as stored."""
    n = 120_000
    idx = _mk(dawn, n, dtype="bf16")
    x = synth.round_bf16(oracle.unit_rows(1, 0, n))
    ids = np.arange(1, n + 1, dtype=np.uint64)
    sc, _ = idx.debug_stream_lists(synth.unit_rows(2, 0, 1)[0])
    assert len(sc) == 256
    for q in np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [n // 3], 4)]):
        for k in (10, 20):
            _assert_same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k))
    assert idx.stats()["fallbacks"] == 0


def test_i6_allocation_failure_falls_back_to_the_int8_stream(dawn, oracle):
    n = 80_000
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    q = synth.planted_queries(1, [5], 4)[0]
    want = oracle.scan_topk(x, ids, q, 10)
    m_all = idx.memory()["shadows"]
    idx.set_option("debug_fail_alloc", 4)  # the 6-bit shadow cannot be allocated: single queries stream the int8 shadow
    assert idx.memory()["shadows"] < m_all
    _assert_same(*idx.search(q, 10), *want)
    idx.add_batch(np.arange(n + 1, n + 101, dtype=np.uint64), synth.unit_rows(3, 0, 100))
    x2 = np.concatenate([x, synth.unit_rows(3, 0, 100)])
    ids2 = np.arange(1, n + 101, dtype=np.uint64)
    _assert_same(*idx.search(q, 10), *oracle.scan_topk(x2, ids2, q, 10))
    idx.set_option("debug_fail_alloc", 0)
    assert idx.memory()["shadows"] >= m_all
    _assert_same(*idx.search(q, 10), *oracle.scan_topk(x2, ids2, q, 10))


def test_i6_behind_a_sharded_handle(dawn, oracle):
    """Every shard of a one-process multi-device handle streams its own 6-bit shadow; the merged answer is the single index's."""
    n = 90_000
    idx = dawn.VectorIndex(0, devices=[0, 0, 0])
    idx.set_option("i6_min_rows", 0)
    idx.fill_synthetic(1, 0, n, 1)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    for q in np.concatenate([synth.unit_rows(2, 0, 2), synth.planted_queries(1, [n // 2], 4)]):
        _assert_same(*idx.search(q, 20), *oracle.scan_topk(x, ids, q, 20))
    assert idx.stats()["fallbacks"] == 0


def test_i6_bits_option_rebuilds_the_shadow(dawn, oracle):
    n = 70_000
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    q = synth.planted_queries(1, [n - 2], 4)[0]
    want = oracle.scan_topk(x, ids, q, 20)
    m = {}
    for b in (6, 5, 6, 5):
        idx.set_option("i6_bits", b)
        m[b] = idx.memory()["shadows"]
        _assert_same(*idx.search(q, 20), *want)
    assert m[6] - m[5] >= n * 48 and idx.stats()["fallbacks"] == 0
    with pytest.raises(Exception):
        idx.set_option("i6_bits", 4)


def test_i6_scan_1m(dawn, oracle):
    n = 1_000_000
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 4), synth.planted_queries(1, [0, 12345, n - 1], 9)])
    for q in Q:
        for k in (10, 64):
            _assert_same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k, threads=8))
    assert idx.stats()["fallbacks"] == 0


@pytest.mark.parametrize("dist", [1, 2, 3])
def test_i6_on_gaussian_and_heavy_tailed_rows(dawn, oracle, dist):
    """Rows with realistic score distributions (option "synth_dist": 1 Gaussian, 2 / 3 a few dimensions x5, as sentence embeddings
    have): the packed stream certifies every query — no exact pass — and returns the oracle's answer for the rows as stored."""
    n = 600_000
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
    idx.set_option("synth_dist", dist)
    idx.fill_synthetic(1, 0, n, 1)
    x, ids = idx.get_rows(0, n)
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    qi.fill_synthetic(2, 0, 6, 1)
    Q = qi.get_rows(0, 6)[0]
    Q[5] = x[4242]
    for q in Q:
        for k in (10, 20):
            _assert_same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k, threads=8))
    assert idx.search(Q[5], 1)[0][0] == 4243
    assert idx.stats()["fallbacks"] == 0


def test_i6_survives_save_load_and_follows_the_int8_switch(dawn, oracle, tmp_path, bits):
    """The packed shadow is rebuilt by `load` like every other shadow; "i8_shadow" = 0 asks for the 16-bit filters: BOTH integer
    shadows stop being read and the packed one's memory goes back (the f16 shadow is built instead, and released again when the
    integer shadows return) — same answers throughout."""
    n = 150_000
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    q = synth.planted_queries(1, [n // 3], 4)[0]
    want = oracle.scan_topk(x, ids, q, 20)
    path = str(tmp_path / "index.dawn")
    idx.save(path)
    other = dawn.VectorIndex(0)
    other.set_option("i6_min_rows", 0)
    other.load(path)
    assert other.size() == n
    _assert_same(*other.search(q, 20), *want)
    sc, _ = other.debug_stream_lists(q)
    assert np.isfinite(other.debug_stream_bound()) and len(sc) == 256  # (the packed stream is what answered)
    rb = 288 if bits == 6 else 240
    m_int = other.memory()["shadows"]
    assert m_int >= n * (384 + rb)
    other.set_option("i8_shadow", 0)
    m_f16 = other.memory()["shadows"]
    assert n * (384 + 768) <= m_f16 < (n + 4096) * (384 + 768 + 8)  # int8 kept, packed released, f16 built
    _assert_same(*other.search(q, 20), *want)
    with pytest.raises(Exception):
        other.debug_stream_bound()  # not live
    other.set_option("i8_shadow", 1)
    assert other.memory()["shadows"] == m_int  # f16 released, packed rebuilt
    _assert_same(*other.search(q, 20), *want)
    assert other.stats()["fallbacks"] == 0


@pytest.mark.parametrize("blocks,threads", [(256, 512), (40, 128), (100, 256), (7, 512), (8, 64), (300, 192)])
def test_i6_dynamic_tail_covers_every_sub_tile(dawn, oracle, blocks, threads):
    """From 16 rounds of the grid on, the last eighth of the index is handed out in strided chunks from shared counters: whatever the
    grid, every sub-tile is scanned exactly once — queries planted all over the dynamic region (and the static one) come back first,
    and the answers equal the oracle's."""
    n = 1_200_003
    idx = _mk(dawn, n)
    idx.set_option("i6_scan_blocks", blocks)
    idx.set_option("i6_scan_threads", threads)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    planted = np.array([0, 31, 500_000, 1_049_999, 1_050_000, 1_100_017, 1_150_000, 1_190_000, n - 40_000, n - 33, n - 1])
    Q = synth.planted_queries(1, planted, 5)
    for r, q in zip(planted, Q):
        lab, dist = idx.search(q, 10)
        assert lab[0] == r + 1, (r, lab)
        _assert_same(lab, dist, *oracle.scan_topk(x, ids, q, 10, threads=8))
    for q in synth.unit_rows(2, 0, 3):  # ... twice: the counters are back at zero after every search
        for _ in range(2):
            _assert_same(*idx.search(q, 20), *oracle.scan_topk(x, ids, q, 20, threads=8))
    sc, rows = idx.debug_stream_lists(Q[3])  # (the stream-only hook resets the counters itself)
    _assert_same(*idx.search(Q[3], 10), *oracle.scan_topk(x, ids, Q[3], 10, threads=8))
    assert idx.stats()["fallbacks"] == 0


# ---- the K2 term of the packed bound, on the sub-tile that stresses it ------------------------------------------------------
def _rot_matrix():
    """The shadows' rotation R (csrc/rotate384.hpp) as a 384 x 384 matrix: x' = R x."""
    k = np.arange(384, dtype=np.uint64)
    neg = ((k * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(19) & np.uint64(1)
    D = np.diag(np.where(neg == 1, -1.0, 1.0))
    M3 = (2.0 / 3.0) * np.ones((3, 3)) - np.eye(3)
    i = np.arange(128)
    H = np.where(np.array([[bin(a & b).count("1") & 1 for b in i] for a in i]) == 1, -1.0, 1.0) / np.sqrt(128.0)
    return np.kron(np.eye(3), H) @ np.kron(M3, np.eye(128)) @ D


def _packed_quantise(xr, levels, n_cand=6):
    """rows_to_i6s_kernel in float64 on rotated rows xr [32, 384]: -> (s, E_true, X)."""
    amax = np.abs(xr).max()
    s0 = np.float32(max(amax, 1e-20)) / np.float32(levels)
    best = None
    for ci in range(n_cand):
        s = float(np.float32(s0 * np.float32(1.0 - np.float32(0.08) * ci)))
        X = np.clip(np.rint(xr / s), -levels, levels)
        e = np.sqrt(((xr - s * X) ** 2).sum(axis=1)).max()
        if best is None or e < best[1]:
            best = (s, e, X)
    return best


def test_k2_covers_the_worst_sub_tile(dawn, oracle, bits):
    """The packed bound is ub = s s_q / 254 C + E + K2 with K2 >= |(s X).dq| through Cauchy-Schwarz: ||s X||_2 ||dq||_2.
    Round 3 took ||s X||_2 <= 1.35 — derived for 31 levels.  Here: a sub-tile built IN THE ROTATED BASIS — a one-hot row (it
    pins the scale at the coarsest value a unit row allows) next to rows whose components all sit just above (n + 1/2) s for
    each of the quantiser's candidate scales, and rows that combine a clipped component with such a body — and a query whose
    int8 images leave the largest residual the format allows (every |dq_i| = 0.49 s_q / 254).  With 15 levels ||s X||_2 reaches
    1.4 whatever scale the quantiser picks, and ||s X||_2 ||dq||_2 exceeds the old constant's K2; the measured form
    (1.015 + E) x 19.6 x 2e-3 x s_q covers it.  Checked through the stream's own lists (option i6_refine = -1: coarse bounds
    kept): K2 as the kernel applied it = ub - the integer term - E >= ||s X_r||_2 ||dq||_2 for every row, and ub >= x.q."""
    levels = 15 if bits == 5 else 31
    R = _rot_matrix()
    assert np.abs(R @ R.T - np.eye(384)).max() < 1e-12
    rng = np.random.default_rng(11)
    amax = 1.0099
    s0 = amax / levels
    rows_r = [np.eye(384)[0] * amax]
    for ci in range(6):
        s = s0 * (1.0 - 0.08 * ci)
        # (a) every component just above a half level: a at (n + 1/2) s, b at (n + 3/2) s, a + b = 384, norm ~ 1
        n0 = int(np.floor(1.0 / (s * np.sqrt(384.0)) - 0.5))
        lo, hi = (n0 + 0.5) ** 2, (n0 + 1.5) ** 2
        b = min(max(int(round((1.0 / s ** 2 - 384.0 * lo) / (hi - lo))), 0), 384)
        mag = np.concatenate([np.full(384 - b, (n0 + 0.5) * s + 0.001 * s), np.full(b, (n0 + 1.5) * s + 0.001 * s)])
        rows_r.append(rng.permutation(mag) * rng.choice([-1.0, 1.0], 384))
        # (b) one large component (clipped by the finer scales) + a body at 0.5 s
        body = 0.5 * s * 1.002
        big = np.sqrt(max(1.0 - 383 * body ** 2, 0.0))
        v = np.full(384, body) * rng.choice([-1.0, 1.0], 384)
        v[0] = min(big, amax * 0.999)
        rows_r.append(v)
    while len(rows_r) < 32:
        v = rng.standard_normal(384)
        rows_r.append(v / np.linalg.norm(v))
    xr = np.stack(rows_r)
    norms = np.linalg.norm(xr, axis=1)
    assert np.all((norms > 0.992) & (norms < 1.00995)), norms
    rows = (xr @ R).astype(np.float32)  # x = R^T x'
    # the query: one component at 127 s_q, the rest H = 0, L random, residual 0.49 / 254 of s_q with the sign of X
    s, e_true, X = _packed_quantise((R @ rows.astype(np.float64).T).T, levels)
    worst = int(np.argmax(np.linalg.norm(s * X, axis=1)))
    sq = 1.0 / np.sqrt(127.0 ** 2 + 383 * (60.0 / 254.0) ** 2)  # ~ 1 / 127.1: ||q'|| ~ 1
    L = rng.integers(-100, 101, 384).astype(np.float64)
    sign = np.where(X[worst] >= 0, 1.0, -1.0)
    qr = sq * (L + 0.49 * sign) / 254.0
    qr[1] = 127.0 * sq  # (component 1: the one-hot row lives in component 0)
    q = (qr @ R).astype(np.float32)
    assert 0.99 < np.linalg.norm(q) < 1.01
    # numpy restatement of the query images (scan_filter_i6s_kernel)
    qrot = R @ q.astype(np.float64)
    sqk = float(np.float32(np.abs(qrot).max()) / np.float32(127.0))
    Hq = np.clip(np.rint(qrot / sqk), -127, 127)
    Lq = np.clip(np.rint((qrot / sqk - Hq) * 254.0), -127, 127)
    dq = qrot - sqk * (Hq + Lq / 254.0)
    assert np.abs(dq).max() < 2.0e-3 * sqk and np.linalg.norm(dq) > 0.9 * np.sqrt(383) * 0.49 / 254.0 * sqk
    sx_norm = np.linalg.norm(s * X, axis=1)
    need = sx_norm * np.linalg.norm(dq)
    k2_old = 1.35 * 19.6 * 2.0e-3 * sqk
    if bits == 5:  # the regime round 3's constant did not cover
        assert sx_norm.max() > 1.38 and need.max() > k2_old, (sx_norm.max(), need.max(), k2_old)
    idx = dawn.VectorIndex(0)
    idx.set_option("i6_min_rows", 0)
    idx.add_batch(np.arange(1, 33, dtype=np.uint64), rows)
    idx.set_option("i6_refine", -1)
    sc, lr = idx.debug_stream_lists(q)
    valid = lr != 0xFFFFFFFF
    got = lr[valid].astype(np.int64)
    assert sorted(got.tolist()) == list(range(32))
    ub = np.zeros(32)
    ub[got] = sc[valid].astype(np.float64)
    Cint = 254.0 * (X @ Hq) + X @ Lq
    e_stored = e_true * 1.0101 * 1.001 + 1e-9
    k2_applied = ub - s * sqk / 254.0 * Cint - e_stored
    exact = rows.astype(np.float64) @ q.astype(np.float64)
    assert np.all(ub >= exact - 4e-6)
    assert np.all(k2_applied >= need - 4e-6), (k2_applied - need).min()
    # ... and the search itself: exact, as always
    idx.set_option("i6_refine", 0)
    _assert_same(*idx.search(q, 10), *oracle.scan_topk(rows, np.arange(1, 33, dtype=np.uint64), q, 10))


@pytest.mark.parametrize("n", [65, 4097, 300_001])
def test_i6_central_tail_and_tail_geometry_options_agree(dawn, oracle, n):
    """Two measured-and-not-adopted variants stay selectable and exact: the central tail (option i6_central_tail: the refined
    lists go to merge_rescore_kernel instead of every workgroup rescoring its own 64 rows — a wash, profiles/r04/
    stream_central_tail_ab.log) and other chunk sizes / shares of the dynamically assigned tail."""
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [n // 2], 4)])
    for opts in ({"i6_central_tail": 1}, {"i6_dyn_chunk": 2, "i6_dyn_share": 6}, {"i6_dyn_chunk": 64, "i6_dyn_share": 15},
                 {"i6_central_tail": 1, "i6_dyn_chunk": 1}):
        for o, v in opts.items():
            idx.set_option(o, v)
        for k in (10, 64):
            for q in Q:
                _assert_same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k))
        idx.set_option("i6_central_tail", 0)
    st = idx.stats()
    assert st["fallbacks"] == 0


def test_list_depth_follows_the_shadows_measured_error_bounds(dawn, oracle, bits):
    """The waves' list depth (scan_i6.hip: i6_refine_count) is sized from the histogram of the shadow's own E (re-read when the shadow
    changes), not from constants: shallower lists on uniform rows, the same answers; one-hot rows next to them (E of their sub-tiles
    is several times larger) move the histogram and the depth with it; "i6_slack_model" = 0 is the round-4 sizing."""
    n = 300_001
    idx = _mk(dawn, n)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    n10, frac = idx.i6_refine(10)
    assert abs(sum(frac) - 1.0) < 1e-4 and n10 == 24, (n10, sum(frac))
    lo = min(b for b, f in enumerate(frac) if f > 0) * 0.004
    mean = sum((b + 0.5) * 0.004 * f for b, f in enumerate(frac))
    # (clipped scales on near-Gaussian rows: E ~0.074 at 5 bits, ~0.037 at 6)
    assert (0.04 < lo and 0.05 < mean < 0.10) if bits == 5 else (0.02 < lo and 0.025 < mean < 0.05), (lo, mean)
    Q = np.concatenate([synth.unit_rows(2, 0, 6), synth.planted_queries(1, [n // 2], 4)])
    for k in (1, 10, 20, 64):
        assert 24 <= idx.i6_refine(k)[0] <= 64
        for q in Q:
            _assert_same(*idx.search(q, k), *oracle.scan_topk(x, ids, q, k))
    idx.set_option("i6_slack_model", 0)
    n10c, fracc = idx.i6_refine(10)
    assert n10c == 40 and sum(fracc) == 0.0
    for q in Q:
        _assert_same(*idx.search(q, 10), *oracle.scan_topk(x, ids, q, 10))
    idx.set_option("i6_slack_model", 1)
    assert idx.i6_refine(10)[0] == 24
    # 4096 one-hot rows: their sub-tiles quantise badly (one component at the scale's end, 383 at zero is exact, but mixed with
    # ordinary rows the sub-tile's scale is the one-hot's) -> the histogram grows a tail, re-read by the add's flush
    hot = np.zeros((4096, 384), dtype=np.float32)
    hot[np.arange(4096), np.arange(4096) % 384] = 1.0
    idx.add_batch(np.arange(n + 1, n + 4097, dtype=np.uint64), hot)
    n10h, frach = idx.i6_refine(10)
    assert abs(sum(frach) - 1.0) < 1e-4 and frach != frac
    x2 = np.concatenate([x, hot])
    ids2 = np.arange(1, n + 4097, dtype=np.uint64)
    for q in list(Q[:3]) + [hot[5]]:
        _assert_same(*idx.search(q, 10), *oracle.scan_topk(x2, ids2, q, 10))
    assert idx.stats()["fallbacks"] == 0
