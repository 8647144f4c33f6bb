"""GPU parity tests for the MiniLM-L6-v2 forward (configs[2] embed leg).

HIP path (dawn_embedder_* through the C ABI) vs the CPU oracle (oracle/dawn_oracle.c restating
src/embedding/model.rs + embedding_service.rs:124-136) and vs the committed golden vectors (HF transformers,
tests/golden/minilm_seed3.npz).  Tolerance (north_star): unit-vector components within 1e-5 f32; hidden states
(|x| = O(1) after LayerNorm) within 5e-5.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL_EMB = 1e-5
TOL_HID = 5e-5


@pytest.fixture(scope="module")
def provider(dawn, tmp_path_factory):
    d = tmp_path_factory.mktemp("model")
    st, cj = dawn.write_synthetic_model(str(d), seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    yield ep
    ep.close()


def test_golden_vectors(provider):
    g = np.load(os.path.join(GOLD, "minilm_seed3.npz"))
    offs, toks = g["seq_offsets"], g["token_ids"]
    seqs = [toks[offs[b]:offs[b + 1]] for b in range(len(offs) - 1)]
    emb = provider.calculate_embedding(seqs)  # one packed batch, mixed lengths 4..128
    assert emb.shape == (len(seqs), 384)
    assert np.abs(emb - g["embeddings"]).max() < TOL_EMB
    assert np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-6
    hs = provider.hidden_states([seqs[0]])[0]
    assert np.abs(hs - g["hidden_states_seq0"]).max() < TOL_HID


def test_batch_equals_batch1_and_oracle(provider, oracle):
    """Every text gets its batch-1 result, whatever it is batched with."""
    sb = oracle.SynthBert(3)
    seqs = synth.token_sequences(21, 9, 2, 40) + [np.array([101, 102], dtype=np.uint32)]
    batch = provider.calculate_embedding(seqs)
    for i, s in enumerate(seqs):
        single = provider.calculate_embedding([s])[0]
        # same per-sequence arithmetic up to the GEMM's k-summation order (a lone short text takes the
        # split-K skinny GEMM, a packed batch the 64x64 tile kernel): rounding-level agreement
        assert np.abs(single - batch[i]).max() < 5e-7
        ref = sb.embed(s)
        assert np.abs(batch[i] - ref).max() < TOL_EMB
    hs = provider.hidden_states(seqs[:3])
    for s, h in zip(seqs[:3], hs):
        assert np.abs(h - sb.forward(s)).max() < TOL_HID


def test_page_lengths_use_the_mfma_attention(provider, oracle):
    """65..128 tokens (pages) take the matrix-core attention kernel; a batch takes the kernel of its longest member, so
    the short texts here run through it as well and must agree with their own batch-1 result (three-phase kernel) and
    with the oracle."""
    sb = oracle.SynthBert(3)
    seqs = [synth.token_sequences(40 + n, 1, n, n)[0] for n in (65, 96, 127, 128, 7, 33, 64)]
    batch = provider.calculate_embedding(seqs)
    for i, s in enumerate(seqs):
        single = provider.calculate_embedding([s])[0]
        assert np.abs(single - batch[i]).max() < 5e-7
        assert np.abs(batch[i] - sb.embed(s)).max() < TOL_EMB
    hs = provider.hidden_states([seqs[2]])[0]
    assert np.abs(hs - sb.forward(seqs[2])).max() < TOL_HID


def test_long_sequences_and_limits(provider, oracle, dawn):
    sb = oracle.SynthBert(3)
    s512 = synth.token_sequences(5, 1, 512, 512)[0]
    assert len(s512) == 512
    e = provider.calculate_embedding([s512, s512[:200]])
    assert np.abs(e[0] - sb.embed(s512)).max() < TOL_EMB
    assert np.abs(e[1] - sb.embed(s512[:200])).max() < TOL_EMB
    with pytest.raises(dawn.DawnError):
        provider.calculate_embedding([np.zeros(513, dtype=np.uint32)])  # > max_position_embeddings
    with pytest.raises(dawn.DawnError):
        provider.calculate_embedding([np.array([101, 40000, 102], dtype=np.uint32)])  # id >= vocab
    with pytest.raises(dawn.DawnError):
        provider.calculate_embedding([np.zeros(0, dtype=np.uint32)])


def test_weight_name_variants(dawn, oracle, tmp_path):
    """'bert.' prefix retry (model.rs:543-547) and LayerNorm gamma/beta fallback (:210-222)."""
    s = synth.token_sequences(3, 1, 12, 12)
    want = oracle.SynthBert(3).embed(s[0])
    for prefix, gb in (("bert.", False), ("", True)):
        st, cj = dawn.write_synthetic_model(str(tmp_path / f"m{len(prefix)}{gb}"), seed=3, prefix=prefix, gamma_beta=gb)
        ep = dawn.EmbeddingProvider(st, cj, 0)
        assert np.abs(ep.calculate_embedding(s)[0] - want).max() < TOL_EMB
        ep.close()
    with pytest.raises(dawn.DawnError):
        dawn.EmbeddingProvider(str(tmp_path / "nope.safetensors"), None, 0)
    cfg = dict(synth.MINILM_CONFIG, hidden_size=768)
    bad = tmp_path / "bad.json"
    bad.write_text(json.dumps(cfg))
    with pytest.raises(dawn.DawnError):
        dawn.EmbeddingProvider(st, str(bad), 0)


def test_embed_then_rank_end_to_end(provider, dawn, oracle):
    """configs[0]/[2] plumbing: pages embedded by the HIP forward, indexed, text query ranked — vs the
    oracle doing the same on the CPU (oracle embed -> oracle scan)."""
    pages = synth.token_sequences(31, 300, 8, 48)
    emb = provider.calculate_embedding(pages)
    ids = np.arange(1, len(pages) + 1, dtype=np.uint64)
    idx = dawn.VectorIndex(0)
    idx.add_batch(ids, emb)
    qs = [pages[17][:9], pages[201]]
    qe = provider.calculate_embedding(qs)
    lab, dist, found = idx.search_batch(qe, 10)
    for b in range(2):
        ol, od = oracle.scan_topk(emb, ids, qe[b], 10)
        assert np.array_equal(lab[b], ol) and np.array_equal(dist[b], od)
    assert lab[1][0] == 202 and dist[1][0] < 1e-6
    # full-CPU chain agrees on the ranking where score gaps exceed the embedding tolerance
    sb = oracle.SynthBert(3)
    cpu_emb = np.stack([sb.embed(p) for p in pages[:60]])
    cl, cd = oracle.scan_topk(cpu_emb, ids[:60], sb.embed(qs[0]), 5)
    idx2 = dawn.VectorIndex(0)
    idx2.add_batch(ids[:60], emb[:60])
    gl, gd = idx2.search(qe[0], 5)
    assert np.abs(gd - cd).max() < 1e-5
    gaps = np.diff(cd)
    if gaps.min() > 1e-4:
        assert np.array_equal(gl, cl)


def test_text_in_vectors_out_with_the_host_tokenizer(provider, dawn, oracle, tmp_path):
    """calculate_embedding(&[&str]) end to end (embedding_service.rs:97-139): C++ WordPiece tokenizer -> packed ids ->
    HIP forward.  The vocabulary is synthetic (ids must stay < 30522); the vectors must equal the oracle's embedding
    of the same ids."""
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"tok{i}" for i in range(95)]
    words += ["dawn", "search", "##es", "the", "web", "distributed", "engine", ",", ".", "a", "##n", "open"]
    # place [CLS]/[SEP] at the BERT ids the synthetic sequences use elsewhere (101 / 102)
    words[2], words[101] = words[101], "[CLS]"
    words[3], words[102] = words[102], "[SEP]"
    vt = tmp_path / "vocab.txt"
    vt.write_text("\n".join(words) + "\n", encoding="utf-8")
    tk = dawn.Tokenizer(str(vt))
    assert tk("Dawn searches the web.") == [101, words.index("dawn"), words.index("search"), words.index("##es"),
                                            words.index("the"), words.index("web"), words.index("."), 102]
    provider.tokenizer = tk
    try:
        texts = ["Dawn searches the web.", "An open, distributed search engine", "unknownword the web"]
        emb = provider.calculate_embedding(texts)
    finally:
        provider.tokenizer = None
    sb = oracle.SynthBert(3)
    for t, e in zip(texts, emb):
        assert np.abs(e - sb.embed(np.array(tk(t), dtype=np.uint32))).max() < TOL_EMB


def test_any_batch_composition_matches_the_oracle(provider, oracle):
    """Property (hypothesis): for any mix of sequence lengths 1..160 in any order, every text gets the oracle's vector —
    the batch's longest member picks the attention kernel (three-phase <= 64, matrix-core <= 128, thread-per-row beyond)
    and its token count picks the GEMM (split-K skinny <= 640 tokens, 64x64 tiles beyond), so the same text travels
    through different kernels depending on its neighbours and must not notice."""
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st

    sb = oracle.SynthBert(3)
    cache = {}

    def ref(seq):
        key = seq.tobytes()
        if key not in cache:
            cache[key] = sb.embed(seq)
        return cache[key]

    pool = [synth.token_sequences(900 + n, 1, n, n)[0] for n in (1, 2, 5, 17, 31, 32, 33, 63, 64, 65, 100, 128, 129, 160)]

    @settings(max_examples=20, deadline=None, suppress_health_check=[HealthCheck.too_slow])
    @given(picks=st.lists(st.integers(0, len(pool) - 1), min_size=1, max_size=14))
    def run(picks):
        seqs = [pool[i] for i in picks]
        emb = provider.calculate_embedding(seqs)
        for s, e in zip(seqs, emb):
            assert np.abs(e - ref(s)).max() < TOL_EMB

    run()


def test_graph_replay_is_bit_identical_to_plain_launches(provider):
    """Launch-bound forwards (a lone text: ~45 kernels of a few microseconds) are replayed as hipGraphs from the second
    sighting of a (batch, tokens, longest sequence) shape on: first call plain, second captured, later ones replayed —
    all bit-identical to the plain-launch result, for changing token contents under one shape and across shapes."""
    shapes = [synth.token_sequences(31 + i, 1, n, n)[0] for i, n in enumerate((2, 5, 17, 32, 33, 64, 100, 128))]
    provider.set_option("graphs", 0)
    plain = [provider.calculate_embedding([s])[0] for s in shapes]
    pair_plain = provider.calculate_embedding([shapes[1], shapes[2]])
    provider.set_option("graphs", 1)
    try:
        for rounds in range(4):  # plain, capture, replay, replay
            for s, ref in zip(shapes, plain):
                got = provider.calculate_embedding([s])[0]
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
            got = provider.calculate_embedding([shapes[1], shapes[2]])
            assert np.array_equal(got.view(np.uint32), pair_plain.view(np.uint32))
        # same shape, other tokens: the graph reads the ids at run time
        other = synth.token_sequences(77, 1, 17, 17)[0]
        provider.set_option("graphs", 0)
        ref = provider.calculate_embedding([other])[0]
        provider.set_option("graphs", 1)
        for _ in range(3):
            assert np.array_equal(provider.calculate_embedding([other])[0].view(np.uint32), ref.view(np.uint32))
    finally:
        provider.set_option("graphs", 1)


def test_second_golden_pin_wide_weights_and_512_tokens(dawn, oracle, tmp_path):
    """HF transformers fixture on the "wide" style-1 weights (bell-shaped, LayerNorm gains 1 +- 0.5, biases 0.1-0.2) with
    sequences of 2 .. 512 tokens in ONE packed batch: the HIP forward within 1e-5 of the fixture and of the C oracle."""
    import json
    g = np.load(os.path.join(GOLD, "minilm_wide_seed5.npz"))
    meta = json.load(open(os.path.join(GOLD, "minilm_wide_seed5.json")))
    st, cj = dawn.write_synthetic_model(str(tmp_path), seed=meta["weight_seed"], style=meta["weight_style"])
    ep = dawn.EmbeddingProvider(st, cj, 0)
    try:
        offs, toks = g["seq_offsets"], g["token_ids"]
        seqs = [toks[offs[b]:offs[b + 1]] for b in range(len(offs) - 1)]
        assert max(len(s) for s in seqs) == 512
        emb = ep.calculate_embedding(seqs)
        assert np.abs(emb - g["embeddings"]).max() < TOL_EMB
        for i in (0, 3, 7):  # each also alone (skinny GEMMs / long-sequence attention)
            assert np.abs(ep.calculate_embedding([seqs[i]])[0] - g["embeddings"][i]).max() < TOL_EMB
        hs = ep.hidden_states([seqs[2]])[0]
        assert np.abs(hs - g["hidden_states_seq2"]).max() < 2e-4  # values up to 5.6: 4e-5 relative
        sb = oracle.SynthBert(meta["weight_seed"], meta["weight_style"])
        assert np.abs(emb[5] - sb.embed(seqs[5])).max() < TOL_EMB
    finally:
        ep.close()


def test_single_kernels_in_isolation(provider, oracle):
    """SURVEY 8(a) rows a3 (BertEmbeddings), a5 (LayerNorm), a7 (tanh-GELU behind the dense) each on its own, against
    float64 numpy restatements of model.rs:266-281, :86-104, :28-37 + :53-64 on the seed-3 weights."""
    w = synth.bert_weights(3)
    T = 37
    ids = synth.token_sequences(77, 1, T, T)[0]

    def ln(v, gam, bet):
        v = v.astype(np.float64)
        m = v.mean(-1, keepdims=True)
        xc = v - m
        var = (xc * xc).mean(-1, keepdims=True)
        return xc / np.sqrt(var + 1e-12) * gam + bet

    # a3: word[id] + type[0] + pos[t] -> LayerNorm
    e = (w["embeddings.word_embeddings.weight"][ids].astype(np.float64) + w["embeddings.token_type_embeddings.weight"][0]
         + w["embeddings.position_embeddings.weight"][:T])
    want = ln(e, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"])
    got = provider.debug_op(0, ids.astype(np.uint32), T)
    assert np.abs(got - want).max() < 2e-6
    # a5: LayerNorm(a + r), inputs of very different scales per row (1e-3 .. 1e3) and a constant row (variance 0 -> eps)
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((T, 384)) * np.logspace(-3, 3, T)[:, None]).astype(np.float32)
    r = (rng.standard_normal((T, 384)) * 0.1).astype(np.float32)
    a[5] = 2.5
    r[5] = -0.5
    want = ln(a.astype(np.float64) + r, w["encoder.layer.0.attention.output.LayerNorm.weight"],
              w["encoder.layer.0.attention.output.LayerNorm.bias"])
    got = provider.debug_op(1, np.concatenate([a, r]), T)
    err = np.abs(got - want)
    assert np.isfinite(got).all() and err[np.arange(T) != 5].max() < 5e-6
    assert np.abs(got[5] - w["encoder.layer.0.attention.output.LayerNorm.bias"]).max() < 1e-6  # (x - mean) = 0 exactly
    # a7 (+ a4): dense 384 -> 1536 + tanh-GELU, inputs spanning the GELU's flat, curved and linear ranges
    x = (rng.standard_normal((T, 384)) * np.linspace(0.05, 6, T)[:, None]).astype(np.float32)
    z = x.astype(np.float64) @ w["encoder.layer.0.intermediate.dense.weight"].astype(np.float64).T \
        + w["encoder.layer.0.intermediate.dense.bias"]
    want = 0.5 * z * (1 + np.tanh(np.sqrt(2 / np.pi) * z * (1 + 0.044715 * z * z)))
    assert np.abs(z).max() > 4 and (np.abs(z) < 0.1).any()
    for op in (2, 3):  # the skinny form and the 64x64 tile kernel
        got = provider.debug_op(op, x, T, out_cols=1536)
        assert np.abs(got - want).max() < 3e-6 * max(1.0, np.abs(want).max())


def test_create_rejects_hostile_config_before_allocating(dawn, tmp_path):
    """Sizes read from config.json are bounded before they size an allocation (a C ABI must not throw through extern "C")."""
    import ctypes as C
    import json
    from dawnsearch_amd import _lib
    st = tmp_path / "model.safetensors"
    st.write_bytes(b"\x02\x00\x00\x00\x00\x00\x00\x00{}")
    for bad in ({"vocab_size": -5}, {"vocab_size": 2 ** 31 - 1}, {"type_vocab_size": 0}, {"intermediate_size": 1 << 30},
                {"max_position_embeddings": 0}, {"layer_norm_eps": -1.0}):
        cj = tmp_path / "config.json"
        cj.write_text(json.dumps({**synth.MINILM_CONFIG, **bad}))
        h = C.c_void_p()
        rc = _lib.lib.dawn_embedder_create(str(st).encode(), str(cj).encode(), 0, C.byref(h))
        assert rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_IO) and not h.value, (bad, rc)
    # a sane config with an empty safetensors file: a clean IO error, not a crash
    (tmp_path / "config.json").write_text(json.dumps(synth.MINILM_CONFIG))
    h = C.c_void_p()
    assert _lib.lib.dawn_embedder_create(str(st).encode(), str(tmp_path / "config.json").encode(), 0, C.byref(h)) == _lib.ERR_IO


def test_wave_attention_f32_output_and_fused_pooling(provider, oracle):
    """The latency form of a call (<= 640 tokens): positions found inside embed_ln, last LayerNorm fused with the pooling
    (<= 64 sequences), and — option "attention_wave" — the wave-per-sequence attention writing f32: all within the bar, for
    one text and for a few, one and two key tiles."""
    sb = oracle.SynthBert(3)
    for wave in (1, 0):
        provider.set_option("attention_wave", wave)
        try:
            for seqs in (synth.token_sequences(11, 1, 27, 27), synth.token_sequences(12, 9, 2, 32), synth.token_sequences(13, 6, 33, 64),
                         synth.token_sequences(14, 70, 2, 8)):  # (70 sequences: the separate pooling launch)
                emb = provider.calculate_embedding(seqs)
                for i, s_ in enumerate(seqs):
                    assert np.abs(emb[i] - sb.embed(s_)).max() < TOL_EMB
                hs = provider.hidden_states(seqs[:2])
                for s_, h in zip(seqs[:2], hs):
                    assert np.abs(h - sb.forward(s_)).max() < TOL_HID
        finally:
            provider.set_option("attention_wave", 0)


def test_register_attention_and_split_ffn_down_of_one_text(provider, oracle):
    """Round 5, one text per call (the reference's only call shape, embedding_service.rs:161-163): attention of sequences of up
    to 32 tokens in registers, one wave per head (attention_regs_kernel: both products transposed so that nothing changes lanes; option
    "attention_wave" = 2 brings the three-phase block kernel back), and the FFN-down layer as four K-slices whose partial sums the next
    LayerNorm adds up (option "ffn2_split"), BertEmbeddings as the prologue of the first layer's Q|K|V launch (option "fused_embed").  Every combination within the bar of the oracle, hidden states and embeddings, for every
    length 1 .. 32, a few texts per call, and 33 .. 64 tokens (two key tiles: the block kernel; still the split FFN-down)."""
    sb = oracle.SynthBert(3)
    cases = [synth.token_sequences(40 + n, 1, n, n) for n in (1, 2, 3, 7, 15, 16, 17, 27, 31, 32)]
    cases += [synth.token_sequences(61, 5, 2, 12), synth.token_sequences(62, 1, 33, 33), synth.token_sequences(63, 1, 64, 64),
              synth.token_sequences(64, 3, 20, 21)]
    base = None
    for aw, split, femb in ((0, 1, 1), (2, 1, 1), (0, 0, 0), (2, 0, 1), (0, 2, 0)):
        provider.set_option("attention_wave", aw)
        provider.set_option("ffn2_split", split)
        provider.set_option("fused_embed", femb)
        try:
            got = []
            for seqs in cases:
                emb = provider.calculate_embedding(seqs)
                for i, s_ in enumerate(seqs):
                    assert np.abs(emb[i] - sb.embed(s_)).max() < TOL_EMB, (aw, split, len(s_))
                hs = provider.hidden_states(seqs[:1])
                assert np.abs(hs[0] - sb.forward(seqs[0])).max() < TOL_HID, (aw, split, len(seqs[0]))
                got.append(emb)
            if base is None:
                base = got
            for a_, b_ in zip(base, got):  # the forms agree far inside the bar
                assert np.abs(np.asarray(a_) - np.asarray(b_)).max() < 2e-6
        finally:
            provider.set_option("attention_wave", 0)
            provider.set_option("ffn2_split", 1)
            provider.set_option("fused_embed", 1)


def test_dense_kernel_forms_are_bit_identical(provider):
    """The 128 x 128 form of the bf16x3 dense kernel has tuning options that change only its schedule — waves of a SIMD in
    lockstep or half a step apart ("gemm3_pingpong"), one workgroup per tile or 256 walking the tile list
    ("gemm3_persistent") — never the arithmetic: a page batch (ragged last tile included) embeds to the same bits."""
    seqs = synth.token_sequences(123, 37, 90, 128)  # ~4000 tokens
    provider.set_option("gemm3_big_min_tiles", 0)
    try:
        ref = provider.calculate_embedding(seqs)
        for pp, pers in ((0, 0), (0, 256), (1, 0), (1, 64)):
            provider.set_option("gemm3_pingpong", pp)
            provider.set_option("gemm3_persistent", pers)
            got = provider.calculate_embedding(seqs)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (pp, pers)
    finally:
        provider.set_option("gemm3_pingpong", 1)
        provider.set_option("gemm3_persistent", 256)
        provider.set_option("gemm3_big_min_tiles", 512)


def test_large_batches_take_the_bf16x3_kernels_and_stay_within_the_bar(provider, oracle):
    """Batches above the skinny limit run their dense layers f32-accurately on the bf16 matrix cores (embed_gemm3.hip: 3-way
    bf16 split, 6 products; 64 x 64 tiles, 128 x 128 tiles from 2048 tokens): same 1e-5 bar against the oracle, and
    rounding-level agreement with the f32-MFMA path (option "gemm_bf16x3" = 0) on the same batch."""
    sb = oracle.SynthBert(3)
    # ~1200 tokens: 64 x 64 tiles; ~3400: 128 x 128 forced; then the wave-per-sequence attention of short-sequence batches
    # with one (<= 32 tokens, down to a single token) and two (33 .. 64) key tiles
    for n_seq, lo, hi, big in ((40, 20, 40, 512), (30, 100, 128, 0), (80, 1, 32, 512), (40, 33, 64, 512)):
        seqs = synth.token_sequences(91 + n_seq, n_seq, lo, hi)
        T = sum(len(s) for s in seqs)
        assert T > 640
        provider.set_option("gemm3_big_min_tiles", big)
        emb = provider.calculate_embedding(seqs)
        for i in (range(n_seq) if hi <= 64 and lo != 20 else (0, n_seq // 2, n_seq - 1)):
            assert np.abs(emb[i] - sb.embed(seqs[i])).max() < TOL_EMB
        provider.set_option("gemm_bf16x3", 0)
        try:
            ref = provider.calculate_embedding(seqs)
        finally:
            provider.set_option("gemm_bf16x3", 1)
        assert np.abs(emb - ref).max() < 2e-6
        hs = provider.hidden_states(seqs[:2])
        for s_, h in zip(seqs[:2], hs):
            assert np.abs(h - sb.forward(s_)).max() < TOL_HID
    provider.set_option("gemm3_big_min_tiles", 0)
    # the dense layer alone, f32 and plane outputs, on 2500 rows (128 x 128 kernel incl. its LDS-staged plane epilogue)
    w = synth.bert_weights(3)
    rng = np.random.default_rng(9)
    T = 2500
    x = (rng.standard_normal((T, 384)) * np.linspace(0.05, 4, T)[:, None]).astype(np.float32)
    z = x.astype(np.float64) @ w["encoder.layer.0.intermediate.dense.weight"].astype(np.float64).T \
        + w["encoder.layer.0.intermediate.dense.bias"]
    want = 0.5 * z * (1 + np.tanh(np.sqrt(2 / np.pi) * z * (1 + 0.044715 * z * z)))
    for op in (3, 4, 5):
        got = provider.debug_op(op, x, T, out_cols=1536)
        assert np.abs(got - want).max() < 3e-6 * max(1.0, np.abs(want).max()), op
    provider.set_option("gemm3_big_min_tiles", 512)
    for T2 in (65, 700):  # the 64 x 64 bf16x3 kernel, ragged last tile
        got = provider.debug_op(5, x[:T2], T2, out_cols=1536)
        assert np.abs(got - want[:T2]).max() < 3e-6 * max(1.0, np.abs(want).max())
