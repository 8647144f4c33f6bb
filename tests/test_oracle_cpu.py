"""CPU suite, part 1: pin the C oracle (oracle/dawn_oracle.c).

The reference has no tests or golden vectors (parity unpinned upstream), so the oracle is pinned against
(a) an independent numpy restatement written from the reference sources (tests/np_oracle.py),
(b) analytical known answers, and (c) golden fixtures generated in the build container from HuggingFace
transformers (tests/golden/, generator tests/golden/make_golden.py) for the embedder.
"""
import json
import os

import numpy as np
import pytest

from dawnsearch_amd import synth
from tests import np_oracle as NP

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_synth_spec_numpy_equals_c(oracle):
    a = oracle.unit_rows(1, 123, 257)
    b = synth.unit_rows(1, 123, 257)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for seed, idx in [(0, 0), (1, 5), (2**40 + 3, 2**50 + 17)]:
        assert oracle.lib().orc_synth_uniform(seed, idx) == synth.uniform(seed, np.array([idx], dtype=np.uint64))[0]
    u = synth.uniform(9, np.arange(200_000, dtype=np.uint64))
    assert -1 < u.min() < -0.999 and 0.999 < u.max() < 1 and abs(u.mean()) < 0.01
    out = np.empty(1000, dtype=np.float32)
    oracle.lib().orc_synth_scaled(3003, 1000, 0.1, 1.0, out)
    assert np.array_equal(out, synth.scaled(3003, 1000, 0.1, 1.0))


def test_vector_functions_vs_numpy(oracle):
    L = oracle.lib()
    X = synth.unit_rows(1, 0, 50)
    q = synth.unit_rows(2, 0, 1)[0]
    d = NP.distances(q, X)
    for r in range(50):
        assert L.orc_distance_cosine(q, X[r]) == d[r]
        assert L.orc_distance_ip(q, X[r]) == NP.seq_dot(q, X[r:r + 1])[0]
        l2 = L.orc_distance_l2sq(q, X[r])
        # unit vectors: sum (a-b)^2 = 2 - 2 a.b = 2 * IP distance
        assert abs(l2 - 2 * d[r]) < 1e-5
        assert L.orc_vector_length(X[r]) == NP.vector_length(X[r])
        assert L.orc_is_normalized(X[r]) == 1
    for scale, ok in [(1.009, True), (1.011, False), (0.991, True), (0.989, False)]:
        v = (X[0] * np.float32(scale)).astype(np.float32)
        assert bool(L.orc_is_normalized(v)) == ok == NP.is_normalized(v)
    v = X[0].copy()
    v[7] = np.inf
    assert L.orc_is_normalized(v) == 0
    v[7] = np.nan
    assert L.orc_is_normalized(v) == 0
    w = (X[1] * np.float32(3.7)).astype(np.float32)
    L.orc_normalize(w, 384)
    assert np.array_equal(w, synth.normalize_rows((X[1] * np.float32(3.7))[None])[0])


def test_i24_codec(oracle):
    L = oracle.lib()
    X = synth.unit_rows(4, 0, 20)
    for v in X:
        enc = np.zeros(1152, dtype=np.uint8)
        L.orc_to24(v, enc)
        assert enc.tobytes() == NP.to24(v)
        dec = np.zeros(384, dtype=np.float32)
        assert L.orc_from24(enc, dec) == 0
        assert np.array_equal(dec, NP.from24(enc.tobytes()))
        assert np.abs(dec - v).max() < 3e-7  # 23-bit grid over [-1,1]
    # known values: -1 -> 0, 0 -> 0x3FFFFF (trunc), 1 -> 0x7FFFFF
    v = np.zeros(384, dtype=np.float32)
    v[0], v[1], v[2] = -1.0, 0.0, 1.0
    enc = np.zeros(1152, dtype=np.uint8)
    L.orc_to24(v, enc)
    assert enc[:9].tolist() == [0, 0, 0, 0xFF, 0xFF, 0x3F, 0xFF, 0xFF, 0x7F]
    # from24 rejects non-unit vectors (vector.rs:70)
    bad = np.zeros(1152, dtype=np.uint8)
    assert L.orc_from24(bad, np.zeros(384, dtype=np.float32)) == -1
    # the "sign extend" branch ORs 0xFF into the LOW byte (vector.rs:65-67), as written
    raw = np.zeros(1152, dtype=np.uint8)
    raw[2] = 0x80
    dec = np.zeros(384, dtype=np.float32)
    L.orc_from24(raw, dec)
    assert dec[0] == np.float32((0x8000FF) / 0x7FFFFF * 2.0 - 1.0)


def test_best_results_semantics(oracle):
    rng = np.random.default_rng(0)
    for size in (1, 3, 20):
        a = oracle.BestResults(size)
        b = NP.BestResults(size)
        assert a.worst_distance() == 0.0  # T::zero() until full (best_results.rs:40)
        for _ in range(300):
            id_ = int(rng.integers(0, 40))
            d = float(np.float32(rng.integers(0, 12) / 4.0))  # many ties
            assert a.insert(id_, d) == b.insert(id_, d)
            assert a.worst_distance() == float(b.worst_distance)
            assert a.results() == [(i, float(x)) for i, x in b.results]
        a.sort()
        b.sort()
        assert a.results() == [(i, float(x)) for i, x in b.results]
    # the tie quirk: the FIRST-positioned worst entry is evicted (best_results.rs:97-107)
    t = oracle.BestResults(2)
    t.insert(0, 5.0)
    t.insert(1, 5.0)
    t.insert(2, 1.0)
    assert t.results() == [(2, 1.0), (1, 5.0)]
    # strict '<': an equal distance never replaces (:56)
    assert t.insert(3, 5.0) is False
    # dedupe by id (:46,:57)
    assert t.insert(2, 0.5) is False


def test_scan_topk_vs_numpy(oracle):
    n = 5000
    X = synth.unit_rows(1, 0, n)
    X[100] = X[7]
    X[4000] = X[7]  # duplicates -> ties -> earlier row first
    ids = (np.arange(n, dtype=np.uint64) * 3 + 11)
    for qi in range(4):
        q = synth.planted_queries(1, [7], 50 + qi)[0] if qi == 0 else synth.unit_rows(2, qi, 1)[0]
        if qi == 0:
            q = X[7].copy()
        for k in (1, 10, 20, 64):
            lab, dist = oracle.scan_topk(X, ids, q, k)
            nlab, ndist = NP.scan_topk(X, ids, q, k)
            assert np.array_equal(lab, nlab) and np.array_equal(dist, ndist)
            lab2, dist2 = oracle.scan_topk(X, ids, q, k, threads=4)
            assert np.array_equal(lab, lab2) and np.array_equal(dist, dist2)
        if qi == 0:
            lab, dist = oracle.scan_topk(X, ids, q, 3)
            assert lab.tolist() == [7 * 3 + 11, 100 * 3 + 11, 4000 * 3 + 11] and dist[0] == dist[2]
    # k > n: found = n
    lab, dist = oracle.scan_topk(X[:5], ids[:5], q, 10)
    assert len(lab) == 5
    lab, dist = oracle.scan_topk(X[:0], ids[:0], q, 10)
    assert len(lab) == 0


def test_examples_old_loop_quirk(oracle):
    """examples_old/search.rs:55-70 compares against results[9] BEFORE the first sort — restated literally.
    After the first accepted replacement the list is sorted and behaves as a plain top-10."""
    n = 400
    X = synth.unit_rows(1, 0, n)
    q = synth.unit_rows(2, 0, 1)[0]
    rec = np.zeros((n, 1568), dtype=np.uint8)
    rec[:, 16:16 + 1536] = X.view(np.uint8).reshape(n, 1536)
    ent = np.zeros(10, dtype=np.uintp)
    sc = np.zeros(10, dtype=np.float32)
    m = oracle.lib().orc_scan_examples_old(rec.reshape(-1), n, q, ent, sc)
    assert m == 10
    # python transcription
    res = []
    for e in range(n):
        s = np.float32(0)
        for i in range(384):
            dlt = np.float32(X[e, i] - q[i])
            s = np.float32(s + np.float32(dlt * dlt))
        if len(res) < 10:
            res.append((s, e))
            continue
        if s < res[9][0]:
            res[9] = (s, e)
            res.sort(key=lambda t: t[0])
    assert [int(x) for x in ent] == [e for _, e in res]
    assert np.array_equal(sc, np.array([s for s, _ in res], dtype=np.float32))


def test_embedder_oracle_vs_golden(oracle):
    """Golden vectors from HF transformers BertModel(gelu_new) on the seed-3 synthetic weights."""
    g = np.load(os.path.join(GOLD, "minilm_seed3.npz"))
    meta = json.load(open(os.path.join(GOLD, "minilm_seed3.json")))
    sb = oracle.SynthBert(meta["weight_seed"])
    offs = g["seq_offsets"]
    toks = g["token_ids"]
    for b in range(len(offs) - 1):
        ids = toks[offs[b]:offs[b + 1]]
        emb = sb.embed(ids)
        assert np.abs(emb - g["embeddings"][b]).max() < 2e-6
        assert abs(np.linalg.norm(emb) - 1) < 1e-6
    hs = sb.forward(toks[offs[0]:offs[1]])
    assert np.abs(hs - g["hidden_states_seq0"]).max() < 2e-5
    # batch-longest padding WITHOUT a mask (embedding_service.rs:101-128) gives a DIFFERENT vector than
    # batch-1 for the shorter texts — the reason the product packs sequences instead of padding
    seqs = [toks[offs[b]:offs[b + 1]] for b in range(3)]
    padded = sb.embed_padded_batch(seqs, pad_id=0)
    lens = [len(s) for s in seqs]
    longest = int(np.argmax(lens))
    for b in range(3):
        same = np.abs(padded[b] - g["embeddings"][b]).max() < 2e-6
        assert same == (lens[b] == lens[longest])


def test_embedder_oracle_vs_second_golden_pin(oracle):
    """Second, independent pin of the embedder restatement: HF transformers on the "wide" style-1 weights (bell-shaped
    values, LayerNorm gains 1 +- 0.5, biases 0.1-0.2: activations up to 5.6) and sequences of 2 .. 512 tokens (512 =
    max_position_embeddings).  tests/golden/minilm_wide_seed5.npz, generator tests/golden/make_golden.py."""
    g = np.load(os.path.join(GOLD, "minilm_wide_seed5.npz"))
    meta = json.load(open(os.path.join(GOLD, "minilm_wide_seed5.json")))
    sb = oracle.SynthBert(meta["weight_seed"], meta["weight_style"])
    offs, toks = g["seq_offsets"], g["token_ids"]
    assert [int(offs[i + 1] - offs[i]) for i in range(len(offs) - 1)] == meta["lengths"] and meta["lengths"][-1] == 512
    for b in range(len(offs) - 1):
        emb = sb.embed(toks[offs[b]:offs[b + 1]])
        assert np.abs(emb - g["embeddings"][b]).max() < 2e-6
    hs = sb.forward(toks[offs[2]:offs[3]])
    assert np.abs(hs - g["hidden_states_seq2"]).max() < 5e-5 and np.abs(g["hidden_states_seq2"]).max() > 3
    # the generators of the style agree between numpy and C bit for bit
    from dawnsearch_amd import synth
    a = np.empty(5000, np.float32)
    oracle.lib().orc_synth_scaled_normal(5003, 5000, 0.5, 1.0, a)
    assert np.array_equal(a, synth.scaled_normal(5003, 5000, 0.5, 1.0))


def test_scan_oracle_vs_heavy_tailed_golden(oracle):
    """Scan fixture on bell-shaped unit rows with four heavy dimensions (what real embeddings look like): the C oracle
    reproduces the numpy restatement's top-20 bit for bit."""
    from dawnsearch_amd import synth
    g = np.load(os.path.join(GOLD, "scan_normal_seed4.npz"))
    n = int(g["n_rows"])
    X = synth.unit_rows_normal(int(g["index_seed"]), 0, n, heavy_dims=tuple(int(d) for d in g["heavy_dims"]))
    assert np.abs(X).max() > 0.4  # heavy-tailed indeed (uniform spec rows: 0.09)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    for q, lab, dist in zip(g["queries"], g["labels"], g["distances"]):
        l, d = oracle.scan_topk(X, ids, q, 20)
        assert np.array_equal(l, lab) and np.array_equal(d.view(np.uint32), dist.view(np.uint32))
    assert g["labels"][3][0] == 1235


def test_topical_mixture_numpy_equals_c_and_its_statistics(oracle):
    """The topical mixture (synth_dist 4 / 5: what the ladder behind the certificates is measured on) is defined by integer
    hashing and single f32 operations: numpy and the C oracle agree bit for bit at any row number; cluster masses follow the
    12-octave Zipf law, the cosine inside a cluster is 1 / (1 + t^2)."""
    from dawnsearch_amd import synth
    for runs in (False, True):
        for first in (0, 99_991, (1 << 40) + 3):
            a = synth.unit_rows_topical(1, first, 300, runs)
            b = oracle.unit_rows_topical(1, first, 300, runs)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert np.all(np.abs(np.linalg.norm(a.astype(np.float64), axis=1) - 1.0) < 1e-6)
    j, t = synth.topical_cluster(1, np.arange(600_000))
    mass = np.bincount(j.astype(np.int64), minlength=4095) / len(j)
    assert abs(mass[0] - 1 / 12) < 3e-3 and abs(mass[1:3].sum() - 1 / 12) < 3e-3 and abs(mass[2047:].sum() - 1 / 12) < 3e-3
    j5, _ = synth.topical_cluster(1, np.arange(4096), runs=True)
    assert np.all(j5.reshape(16, 256) == j5.reshape(16, 256)[:, :1])  # 256 consecutive rows share a cluster
    x = oracle.unit_rows_topical(1, 0, 60_000)
    for c in (0, 1):
        m = np.nonzero(j[:60_000] == c)[0][:300]
        g = x[m].astype(np.float64) @ x[m].astype(np.float64).T
        want = 1.0 / (1.0 + float(t[m[0]]) ** 2)
        assert abs(g[np.triu_indices(len(m), 1)].mean() - want) < 0.03
    # the oracle's scan of rows generated on the fly = its scan of the same rows materialised = the numpy restatement's
    n = 20_000
    X = oracle.unit_rows_topical(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    Q = synth.unit_rows_topical(1, 1 << 40, 3)
    lab, dist = oracle.scan_topk_synth(1, 0, n, 1, Q, 10, dist=4)
    for b in range(3):
        l1, d1 = oracle.scan_topk(X, ids, Q[b], 10)
        assert np.array_equal(lab[b], l1) and np.array_equal(dist[b].view(np.uint32), d1.view(np.uint32))
        l2, d2 = NP.scan_topk(X, ids, Q[b], 10)
        assert np.array_equal(l1, l2) and np.array_equal(d1.view(np.uint32), np.asarray(d2, dtype=np.float32).view(np.uint32))
