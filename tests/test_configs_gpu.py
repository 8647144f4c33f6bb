"""BASELINE.json's configurations, each run AS CONFIGURED on one GPU and checked against the CPU oracle chain.

  configs[0]  10 k synthetic pages embedded, one text query, brute-force top-10 (the plumbing case)
  configs[2]  1 M x 384 index, 256 TEXT queries: host tokenizer -> MiniLM HIP forward -> cosine scan, end to end
  configs[3]  one shard of the 100 M index on 8 GPUs: 12.5 M rows, batch 256 (and batch 1)
(configs[1] — 1 M rows, one query vector — is test_scan_gpu.py; the 100 M-row index itself and the 125 M-row bf16 shard of
configs[4] are checked against the oracle in test_full_size_gpu.py.)

The oracle side: `oracle.scan_topk_synth` scans synthetic rows it generates chunk by chunk (dawn_oracle.c:
orc_scan_topk_synth = orc_synth_unit_row + orc_distance_cosine + the (distance, position) top-k), `oracle.SynthBert` is the C
restatement of the reference's forward.  Bars: scans bit-identical (labels and distance bits); embeddings within 1e-5;
rankings of the two full chains equal wherever the score gaps exceed the embedding tolerance.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402

TOL_EMB = 1e-5


@pytest.fixture(scope="module")
def provider(dawn, tmp_path_factory):
    d = tmp_path_factory.mktemp("model")
    st, cj = dawn.write_synthetic_model(str(d), seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    yield ep
    ep.close()


def _vocab_file(tmp_path, n_words=3000):
    """A synthetic WordPiece vocabulary with [CLS] / [SEP] at BERT's ids 101 / 102 and `n_words` whole words + suffixes."""
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"unused{i}" for i in range(96)] + ["x101", "x102"]
    words[2], words[101] = words[101], "[CLS]"
    words[3], words[102] = words[102], "[SEP]"
    syll = ["ka", "lo", "mi", "ne", "su", "ta", "ri", "vo", "de", "pu", "sha", "qui", "zor", "bel", "fen", "gar"]
    stems = []
    for a in syll:
        for b in syll:
            for c in syll[:12]:
                stems.append(a + b + c)
    stems = stems[:n_words]
    words += stems + ["##" + s for s in syll] + [",", ".", "?", "!", "the", "of", "and", "a"]
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(words) + "\n", encoding="utf-8")
    return str(p), stems, syll


def _texts(seed, n, stems, syll, lo, hi):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ws = []
        for _ in range(int(rng.integers(lo, hi + 1))):
            w = stems[int(rng.integers(len(stems)))]
            r = rng.random()
            if r < 0.15:
                w = w + syll[int(rng.integers(len(syll)))]          # stem + ##suffix
            elif r < 0.2:
                w = w.capitalize()                                   # lowercased by the normaliser
            elif r < 0.23:
                w = "zzzqqq" + w                                     # no such piece: [UNK]
            ws.append(w)
            if rng.random() < 0.1:
                ws.append(str(rng.choice([",", ".", "?", "the", "of"])))
        out.append(" ".join(ws))
    return out


def _oracle_embed_many(oracle, seqs, threads=16):
    """The C oracle's forward for many sequences (one text per call, as the reference), host threads in parallel (the C
    call releases the GIL; each thread owns its SynthBert handle)."""
    chunks = [list(range(i, len(seqs), threads)) for i in range(threads)]

    def work(ix):
        sb = oracle.SynthBert(3)
        return [(i, sb.embed(np.asarray(seqs[i], dtype=np.uint32))) for i in ix]

    out = [None] * len(seqs)
    with ThreadPoolExecutor(threads) as ex:
        for part in ex.map(work, chunks):
            for i, v in part:
                out[i] = v
    return np.stack(out)


def test_configs2_256_texts_tokenizer_forward_scan_1m(dawn, oracle, provider, tmp_path):
    """configs[2] as one composite: 256 texts -> dawn_tokenizer_encode_batch -> dawn_embedder_forward -> top-10 over 1 M
    rows (embedding_service.rs:97-139 then examples_old/search.rs:44-72), against tokenizer ids -> oracle forward -> oracle
    scan."""
    n, B, k = 1_000_000, 256, 10
    vocab, stems, syll = _vocab_file(tmp_path)
    tk = dawn.Tokenizer(vocab)
    texts = _texts(11, B, stems, syll, 2, 22)
    flat, offs = tk.encode_batch(texts)
    seqs = [flat[offs[b]:offs[b + 1]] for b in range(B)]
    assert all(s[0] == 101 and s[-1] == 102 and 3 <= len(s) <= 128 for s in seqs)
    assert sum(1 in s.tolist() for s in seqs) > 0  # some [UNK]s travel through
    for b in (0, 100, 255):  # batch encoding = one-by-one encoding
        assert np.array_equal(tk.encode(texts[b]), seqs[b])
    provider.tokenizer = tk
    try:
        emb = provider.calculate_embedding(texts)  # text in, vectors out: ONE packed batch of 256
    finally:
        provider.tokenizer = None
    assert emb.shape == (B, 384)
    ref = _oracle_embed_many(oracle, seqs)
    assert np.abs(emb - ref).max() < TOL_EMB
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    labels, dist, found = idx.search_batch(emb, k)  # all 256 in one call: the matrix-core pass
    assert np.all(found == k)
    # (1) the scan: bit-equal to the oracle scan of the SAME vectors
    ol, od = oracle.scan_topk_synth(1, 0, n, 1, emb, k)
    assert np.array_equal(labels, ol) and np.array_equal(dist.view(np.uint32), od.view(np.uint32))
    # (2) the whole chain against the whole oracle chain (oracle vectors -> oracle scan): scores within the embedding
    # tolerance, identical ranking wherever neighbouring scores are further apart than that
    cl, cd = oracle.scan_topk_synth(1, 0, n, 1, ref, k + 1)
    assert np.abs(dist - cd[:, :k]).max() < 2e-5
    agree = 0
    for b in range(B):
        gaps = np.diff(cd[b])
        if gaps.min() > 4e-5:
            assert np.array_equal(labels[b], cl[b, :k]), b
            agree += 1
    assert agree > B // 2
    # one text per call (the reference's only call shape) gives the same answer as its slot in the batch
    provider.tokenizer = tk
    try:
        for b in (3, 77):
            e1 = provider.calculate_embedding([texts[b]])[0]
            assert np.abs(e1 - emb[b]).max() < 5e-7
            l1, d1 = idx.search(e1, k)
            o1 = oracle.scan_topk_synth(1, 0, n, 1, e1, k)
            assert np.array_equal(l1, o1[0][0]) and np.array_equal(d1.view(np.uint32), o1[1][0].view(np.uint32))
    finally:
        provider.tokenizer = None


def test_configs0_10k_pages_single_text_query(dawn, oracle, provider, tmp_path):
    """configs[0] at its size: 10 k pages (token sequences of 16..128 ids) embedded by the HIP forward in packed batches,
    inserted through SearchProvider (search_provider.rs:250-286), one TEXT query -> top-10.  Oracle: the C forward on a
    sample of the pages and on the query; the exact scan (and the examples_old L2^2 loop, search.rs:49-72) over the index's
    own 10 k vectors."""
    from dawnsearch_amd.search_provider import ExtractedPage
    n = 10_000
    pages = synth.token_sequences(41, n, 16, 128)
    emb = np.concatenate([provider.calculate_embedding(pages[i:i + 500]) for i in range(0, n, 500)])
    assert emb.shape == (n, 384) and np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-6
    sample = list(range(0, n, 417))  # 24 pages through the CPU oracle forward
    ref = _oracle_embed_many(oracle, [pages[i] for i in sample])
    assert np.abs(emb[sample] - ref).max() < TOL_EMB
    sp = dawn.SearchProvider(0)
    for i in range(0, n, 1000):  # the provider's bulk insert; ids = 1-based rowids (:275-277)
        sp.insert_batch([ExtractedPage(url=f"https://example.org/{j}", title=f"t{j}") for j in range(i, i + 1000)], emb[i:i + 1000])
    assert sp.page_count() == n and sp.index.size() == n
    vocab, stems, syll = _vocab_file(tmp_path)
    tk = dawn.Tokenizer(vocab)
    provider.tokenizer = tk
    try:
        text = " ".join(stems[7:19])
        q = provider.calculate_embedding([text])[0]
    finally:
        provider.tokenizer = None
    assert np.abs(q - oracle.SynthBert(3).embed(tk.encode(text))).max() < TOL_EMB
    res = sp.search_embedding(q)  # count = 20 (search_provider.rs:214)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    ol, od = oracle.scan_topk(emb, ids, q, 20)
    assert [p.page_id for p in res.pages] == ol.tolist()
    assert np.array_equal(np.array([p.distance for p in res.pages], dtype=np.float32).view(np.uint32), od.view(np.uint32))
    assert res.pages_searched == n and res.pages[0].url == f"https://example.org/{int(ol[0]) - 1}"
    lab, dist = sp.index.search(q, 10)  # the configuration's top-10
    assert np.array_equal(lab, ol[:10]) and np.array_equal(dist.view(np.uint32), od[:10].view(np.uint32))
    # the reference's own brute-force loop ranks by L2^2 over PageEntry records: same order for unit vectors
    rec = np.zeros((n, 1568), dtype=np.uint8)
    rec[:, 16:16 + 1536] = emb.view(np.uint8).reshape(n, 1536)
    ent = np.zeros(10, dtype=np.uintp)
    sc = np.zeros(10, dtype=np.float32)
    got = oracle.lib().orc_scan_examples_old(rec.reshape(-1), n, q, ent, sc)
    assert got == 10
    gaps = np.diff(od[:11])
    if gaps.min() > 1e-6:
        assert (ent + 1).tolist() == ol[:10].tolist()
    # a page queried by its own vector comes back first at distance ~0 (search_like, :194-200)
    r2 = sp.search_like(4242)
    assert r2.pages[0].page_id == 4242 and r2.pages[0].distance < 1e-6


def test_configs3_shard_12p5m_rows_batch256_vs_oracle(dawn, oracle):
    """configs[3]: 100 M rows over 8 GPUs = 12.5 M rows per GPU, batch 256, k = 20 (the service's count).  One such shard
    (rows 37.5 M .. 50 M of the index: shard 3 of 8 under contiguous sharding, labels follow the rows): all 256 queries in
    one call, a sample of them against the oracle scan of the same 12.5 M rows; batch-1 on the same shard."""
    n, first = 12_500_000, 37_500_000
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, first, n, first + 1)
    Q = synth.unit_rows(2, 0, 256)
    planted = np.array([first, first + 6_000_000, first + n - 1])
    Q[:3] = synth.planted_queries(1, planted, 7)
    labels, dist, found = idx.search_batch(Q, 20)
    assert np.all(found == 20) and labels.min() >= first + 1 and labels.max() <= first + n
    assert np.array_equal(labels[:3, 0], planted + 1)
    sample = [0, 1, 2, 3, 64, 129, 200, 255]
    ol, od = oracle.scan_topk_synth(1, first, n, first + 1, Q[sample], 20)
    assert np.array_equal(labels[sample], ol) and np.array_equal(dist[sample].view(np.uint32), od.view(np.uint32))
    assert idx.memory()["shadows"] > n * (384 + 240)  # a shard of this size keeps the packed 5-bit shadow for its single queries
    for j, b in enumerate(sample[:4]):  # the same queries one at a time: the packed stream; k = 10
        l1, d1 = idx.search(Q[b], 10)
        assert np.array_equal(l1, ol[j, :10]) and np.array_equal(d1.view(np.uint32), od[j, :10].view(np.uint32))
    idx.set_option("i6_shadow", 0)  # ... and the int8 stream
    for j, b in enumerate(sample[:2]):
        l1, d1 = idx.search(Q[b], 20)
        assert np.array_equal(l1, ol[j]) and np.array_equal(d1.view(np.uint32), od[j].view(np.uint32))
    assert idx.stats()["fallbacks"] == 0
    idx.close()
