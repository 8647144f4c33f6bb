"""Synthetic data spec in numpy — deterministic, counter-based, bit-identical to the
CPU checker's C implementation and to the on-GPU generator (`csrc/scan_kernels.hip: synth_*_kernel`).

Nothing here restates reference behaviour; it only defines the seeded inputs that the parity tests
and bench.py feed to both the HIP path and the oracle.

  uniform(seed, i) = (2*u + 1 - 2^24) / 2^24,  u = top 24 bits of
                     splitmix64(splitmix64(seed) + i * 0x9E3779B97F4A7C15)
  unit row r of stream `seed` = normalise([uniform(seed, r*384 + c) for c in 0..384]) with the
                     reference's sequential f32 sum / sqrt / divide (src/search/vector.rs:194-197)
"""
from __future__ import annotations

import numpy as np

EM_LEN = 384
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, idx: np.ndarray) -> np.ndarray:
    """24-bit uniform in (-1, 1), exactly representable in f32."""
    idx = np.asarray(idx, dtype=np.uint64)
    key = splitmix64(np.uint64(seed))
    with np.errstate(over="ignore"):
        h = splitmix64(key + idx * _GOLDEN)
    u = (h >> np.uint64(40)).astype(np.int64)
    n = 2 * u + 1 - (1 << 24)
    return (n.astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def _seq_sum_sq(rows: np.ndarray) -> np.ndarray:
    """Sequential (left-to-right) f32 sum of squares per row — vector.rs:195 order."""
    s = np.zeros(rows.shape[0], dtype=np.float32)
    for c in range(rows.shape[1]):
        s = (s + rows[:, c] * rows[:, c]).astype(np.float32)
    return s


def normalize_rows(rows: np.ndarray) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    length = np.sqrt(_seq_sum_sq(rows)).astype(np.float32)
    return (rows / length[:, None]).astype(np.float32)


def unit_rows(seed: int, first_row: int, n: int) -> np.ndarray:
    """[n, 384] f32 unit rows of stream `seed` starting at row `first_row`."""
    idx = (np.uint64(first_row) * np.uint64(EM_LEN)
           + np.arange(n * EM_LEN, dtype=np.uint64)).reshape(n, EM_LEN)
    return normalize_rows(uniform(seed, idx))


def unit_rows_normal(seed: int, first_row: int, n: int, heavy_dims=(), heavy_scale: float = 5.0) -> np.ndarray:
    """[n, 384] unit rows with bell-shaped components (scaled_normal's g over stream `seed`), optionally with the
    dimensions `heavy_dims` scaled up before normalisation — the score distribution of real embeddings, and what the
    int8 shadow's quantiser has to cope with (tests; identical values wherever numpy runs)."""
    g = scaled_normal(seed, (first_row + n) * EM_LEN, 1.0, 0.0)[first_row * EM_LEN:].reshape(n, EM_LEN).copy()
    for d in heavy_dims:
        g[:, d] = (g[:, d] * np.float32(heavy_scale)).astype(np.float32)
    return normalize_rows(g)


# ---- topical mixture ("synth_dist" 4 / 5): what a crawl's page vectors look like to the filters ----
# Zipf-sized clusters around bell-shaped centroids; a row = normalise(centroid + t * noise), t per cluster so that the
# cosine between two pages of one topic is 0.5 ... 0.95.  Integer hashing and f32 adds / multiplies in a fixed order only:
# bit-identical in numpy, the C oracle (orc_synth_topical_row) and the GPU generator (scan_kernels.hip: synth_value<4 / 5>).
#   unit(r)    = r (dist 4: topics interleaved row by row) or r >> 8 (dist 5: runs of 256 consecutive rows share a topic — the
#                pages of one site are inserted back to back, src/index/warc.rs:75-86, search_provider.rs:250-286)
#   h          = splitmix64(key ^ unit * 0xD1B54A32D192ED03 ^ TOPIC_SALT),  key = splitmix64(seed)
#   octave o   = (h >> 32) % 12;  cluster j = 2^o - 1 + (h & (2^o - 1))  in [0, 4095): mass 1 / (12 * 2^o) ~ Zipf(1)
#   t_j        = TOPIC_T[(splitmix64(key ^ j * 0xD6E8FEB86659FD93 ^ LEVEL_SALT) >> 20) % 6]   (cosine .5 .6 .7 .8 .9 .95)
#   centroid   = g4(splitmix64(key ^ CENTROID_SALT); j * 384 + col),  noise = g4(key; r * 384 + col)
#   g4(k; i)   = (((u(4i) + u(4i+1)) + u(4i+2)) + u(4i+3)) * 0.8660254   (unit variance, |g| <= 3.47)
#   value      = centroid + t_j * noise, then the reference's sequential-sum normalisation
TOPIC_SALT = np.uint64(0x746F706963730001)
LEVEL_SALT = np.uint64(0x746F706963730002)
CENTROID_SALT = np.uint64(0x746F706963730003)
TOPIC_MUL = np.uint64(0xD1B54A32D192ED03)
LEVEL_MUL = np.uint64(0xD6E8FEB86659FD93)
TOPIC_OCTAVES = 12
TOPIC_T = np.array([1.0, 0.8164966, 0.6546537, 0.5, 0.33333334, 0.22941573], dtype=np.float32)
TOPIC_RUN_SHIFT = 8


def _uniform_key(key: np.uint64, idx: np.ndarray) -> np.ndarray:
    """uniform() on an already-hashed stream key."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(key + idx * _GOLDEN)
    u = (h >> np.uint64(40)).astype(np.int64)
    n = 2 * u + 1 - (1 << 24)
    return (n.astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def _g4_key(key: np.uint64, i: np.ndarray) -> np.ndarray:
    i4 = np.asarray(i, dtype=np.uint64) * np.uint64(4)
    u0, u1, u2, u3 = (_uniform_key(key, i4 + np.uint64(j)) for j in range(4))
    g = (((u0 + u1).astype(np.float32) + u2).astype(np.float32) + u3).astype(np.float32)
    return (g * np.float32(0.8660254)).astype(np.float32)


def topical_cluster(seed: int, rows: np.ndarray, runs: bool = False):
    """(cluster id, noise level t) of the rows `rows` (absolute row numbers) of stream `seed`."""
    rows = np.asarray(rows, dtype=np.uint64)
    key = splitmix64(np.uint64(seed))
    unit = rows >> np.uint64(TOPIC_RUN_SHIFT) if runs else rows
    with np.errstate(over="ignore"):
        h = splitmix64(key ^ (unit * TOPIC_MUL) ^ TOPIC_SALT)
        o = (h >> np.uint64(32)) % np.uint64(TOPIC_OCTAVES)
        one = np.uint64(1)
        j = ((one << o) - one) + (h & ((one << o) - one))
        hj = splitmix64(key ^ (j * LEVEL_MUL) ^ LEVEL_SALT)
    t = TOPIC_T[((hj >> np.uint64(20)) % np.uint64(len(TOPIC_T))).astype(np.int64)]
    return j, t


def unit_rows_topical(seed: int, first_row: int, n: int, runs: bool = False) -> np.ndarray:
    """[n, 384] unit rows of the topical mixture (synth_dist 4; runs = True: 5) starting at row `first_row`."""
    rows = np.uint64(first_row) + np.arange(n, dtype=np.uint64)
    key = splitmix64(np.uint64(seed))
    j, t = topical_cluster(seed, rows, runs)
    cols = np.arange(EM_LEN, dtype=np.uint64)[None, :]
    cen = _g4_key(splitmix64(key ^ CENTROID_SALT), j[:, None] * np.uint64(EM_LEN) + cols)
    noi = _g4_key(key, rows[:, None] * np.uint64(EM_LEN) + cols)
    v = (cen + (t[:, None] * noi).astype(np.float32)).astype(np.float32)
    return normalize_rows(v)


def round_bf16(a: np.ndarray) -> np.ndarray:
    """f32 -> nearest-even bf16 -> f32 (what a DAWN_DTYPE_BF16 index stores and scores)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def scaled(seed: int, n: int, scale: float, offset: float) -> np.ndarray:
    u = uniform(seed, np.arange(n, dtype=np.uint64))
    m = (np.float32(scale) * u).astype(np.float32)
    return (np.float32(offset) + m).astype(np.float32)


def planted_queries(index_seed: int, rows: np.ndarray, noise_seed: int, noise: float = 0.05) -> np.ndarray:
    """Queries whose nearest neighbour is known: query_i = normalise(row[rows[i]] + noise * u)."""
    out = []
    for i, r in enumerate(np.asarray(rows, dtype=np.int64)):
        base = unit_rows(index_seed, int(r), 1)[0]
        u = uniform(noise_seed, np.uint64(i) * np.uint64(EM_LEN) + np.arange(EM_LEN, dtype=np.uint64))
        out.append((base + np.float32(noise) * np.float32(1.0 / np.sqrt(EM_LEN)) * u).astype(np.float32))
    return normalize_rows(np.stack(out))


# ---- synthetic all-MiniLM-L6-v2-shaped weights (tensor list documented in DESIGN.md §5) ----

MINILM_CONFIG = dict(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                     intermediate_size=1536, hidden_act="gelu", hidden_dropout_prob=0.1,
                     max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02,
                     layer_norm_eps=1e-12, pad_token_id=0, model_type="bert")


def bert_tensor_specs(cfg: dict = MINILM_CONFIG):
    """[(name, shape, scale, offset)] in stream order (tensor t uses seed*1000 + t)."""
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    specs = [
        ("embeddings.word_embeddings.weight", (cfg["vocab_size"], H), 0.05, 0.0),
        ("embeddings.position_embeddings.weight", (cfg["max_position_embeddings"], H), 0.02, 0.0),
        ("embeddings.token_type_embeddings.weight", (cfg["type_vocab_size"], H), 0.02, 0.0),
        ("embeddings.LayerNorm.weight", (H,), 0.10, 1.0),
        ("embeddings.LayerNorm.bias", (H,), 0.05, 0.0),
    ]
    for L in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{L}."
        specs += [
            (p + "attention.self.query.weight", (H, H), 0.08, 0.0),
            (p + "attention.self.query.bias", (H,), 0.02, 0.0),
            (p + "attention.self.key.weight", (H, H), 0.08, 0.0),
            (p + "attention.self.key.bias", (H,), 0.02, 0.0),
            (p + "attention.self.value.weight", (H, H), 0.08, 0.0),
            (p + "attention.self.value.bias", (H,), 0.02, 0.0),
            (p + "attention.output.dense.weight", (H, H), 0.05, 0.0),
            (p + "attention.output.dense.bias", (H,), 0.02, 0.0),
            (p + "attention.output.LayerNorm.weight", (H,), 0.10, 1.0),
            (p + "attention.output.LayerNorm.bias", (H,), 0.05, 0.0),
            (p + "intermediate.dense.weight", (I, H), 0.05, 0.0),
            (p + "intermediate.dense.bias", (I,), 0.02, 0.0),
            (p + "output.dense.weight", (H, I), 0.03, 0.0),
            (p + "output.dense.bias", (H,), 0.02, 0.0),
            (p + "output.LayerNorm.weight", (H,), 0.10, 1.0),
            (p + "output.LayerNorm.bias", (H,), 0.05, 0.0),
        ]
    return specs


def scaled_normal(seed: int, n: int, scale: float, offset: float) -> np.ndarray:
    """offset + scale * g, g = ((u0 + u1) + u2 + u3) * sqrt(3)/2 over four consecutive uniforms of the stream: a
    bell-shaped value of unit variance (|g| <= 3.46) with f32 arithmetic in a fixed order (identical in numpy and C)."""
    i = np.arange(n, dtype=np.uint64) * np.uint64(4)
    u0, u1, u2, u3 = (uniform(seed, i + np.uint64(j)) for j in range(4))
    g = (((u0 + u1).astype(np.float32) + u2).astype(np.float32) + u3).astype(np.float32)
    g = (g * np.float32(0.8660254)).astype(np.float32)
    m = (np.float32(scale) * g).astype(np.float32)
    return (np.float32(offset) + m).astype(np.float32)


# style 1 ("wide"): bell-shaped weights, LayerNorm gains 1 +- 0.5 (some near 0, some beyond 2), biases of 0.1-0.2 —
# further from the uniform style-0 weights than a trained checkpoint is: a second, independent pin of the embedder
_WIDE = {"word": 0.06, "pos": 0.03, "type": 0.03, "ln_g": 0.5, "ln_b": 0.2, "qkv_w": 0.06, "bias": 0.1, "ao_w": 0.05,
         "i_w": 0.04, "o_w": 0.03}


def bert_tensor_specs_wide(cfg: dict = MINILM_CONFIG):
    H, I = cfg["hidden_size"], cfg["intermediate_size"]
    W = _WIDE
    specs = [
        ("embeddings.word_embeddings.weight", (cfg["vocab_size"], H), W["word"], 0.0),
        ("embeddings.position_embeddings.weight", (cfg["max_position_embeddings"], H), W["pos"], 0.0),
        ("embeddings.token_type_embeddings.weight", (cfg["type_vocab_size"], H), W["type"], 0.0),
        ("embeddings.LayerNorm.weight", (H,), W["ln_g"], 1.0),
        ("embeddings.LayerNorm.bias", (H,), W["ln_b"], 0.0),
    ]
    for L in range(cfg["num_hidden_layers"]):
        p = f"encoder.layer.{L}."
        specs += [
            (p + "attention.self.query.weight", (H, H), W["qkv_w"], 0.0),
            (p + "attention.self.query.bias", (H,), W["bias"], 0.0),
            (p + "attention.self.key.weight", (H, H), W["qkv_w"], 0.0),
            (p + "attention.self.key.bias", (H,), W["bias"], 0.0),
            (p + "attention.self.value.weight", (H, H), W["qkv_w"], 0.0),
            (p + "attention.self.value.bias", (H,), W["bias"], 0.0),
            (p + "attention.output.dense.weight", (H, H), W["ao_w"], 0.0),
            (p + "attention.output.dense.bias", (H,), W["bias"], 0.0),
            (p + "attention.output.LayerNorm.weight", (H,), W["ln_g"], 1.0),
            (p + "attention.output.LayerNorm.bias", (H,), W["ln_b"], 0.0),
            (p + "intermediate.dense.weight", (I, H), W["i_w"], 0.0),
            (p + "intermediate.dense.bias", (I,), W["bias"], 0.0),
            (p + "output.dense.weight", (H, I), W["o_w"], 0.0),
            (p + "output.dense.bias", (H,), W["bias"], 0.0),
            (p + "output.LayerNorm.weight", (H,), W["ln_g"], 1.0),
            (p + "output.LayerNorm.bias", (H,), W["ln_b"], 0.0),
        ]
    return specs


def bert_weights(seed: int, cfg: dict = MINILM_CONFIG, style: int = 0) -> dict:
    """style 0: uniform values (the DESIGN.md §5 spec); style 1: the "wide" bell-shaped weights above."""
    out = {}
    specs = bert_tensor_specs(cfg) if style == 0 else bert_tensor_specs_wide(cfg)
    for t, (name, shape, scale, offset) in enumerate(specs):
        n = int(np.prod(shape))
        gen = scaled if style == 0 else scaled_normal
        out[name] = gen(seed * 1000 + t, n, scale, offset).reshape(shape)
    return out


def token_sequences(seed: int, B: int, min_len: int = 4, max_len: int = 32):
    """Synthetic WordPiece id sequences: [CLS]=101, ids uniform in [1000, 30521], [SEP]=102."""
    seqs = []
    key_len = splitmix64(np.uint64(seed))
    key_tok = splitmix64(np.uint64(seed + 7919))
    with np.errstate(over="ignore"):
        for b in range(B):
            h = splitmix64(key_len + np.uint64(b) * _GOLDEN)
            L = min_len + int(h % np.uint64(max_len - min_len + 1))
            body_n = max(L - 2, 0)
            ctr = np.uint64(b) * np.uint64(1024) + np.arange(body_n, dtype=np.uint64)
            hh = splitmix64(key_tok + ctr * _GOLDEN)
            body = (np.uint64(1000) + hh % np.uint64(30522 - 1000)).astype(np.uint32)
            seqs.append(np.concatenate([[101], body, [102]]).astype(np.uint32)[:max(L, 2)])
    return seqs
