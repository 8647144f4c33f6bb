"""Property tests (hypothesis) of the whole search path against the CPU oracle: for ANY index content the GPU answer is
the oracle's, bit for bit — the certificate either proves the filtered shortlist or hands the query to the exact pass.
The generated indexes are built to stress exactly that machinery: exact duplicates (ties -> earlier row), near-ties
below every filter bound, tight clusters around the query (more than 64 rows inside the filter's error band), sparse
vectors with tiny components (f16 subnormal territory), antipodal and orthogonal rows, batches that mix all of them,
rows that stretch the int8 quantiser (one-hot / two-hot rows that set the scale of their whole sub-tile, Gaussian rows),
any k up to 64, both index types, every filter source (packed 5- / 6-bit shadow forced on these small indexes — single queries —,
int8 upper-bound shadow, 16-bit rows), every batch size class (stream 1..3, matrix-core 4+, forced 8-per-pass stream)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.sqrt((v * v).sum())).astype(np.float32)


def _build_rows(rng, n, kind, q0):
    base = synth.unit_rows(int(rng.integers(1, 1 << 30)), 0, n)
    if kind == "random" or n < 4:
        return base
    rows = base.copy()
    m = int(rng.integers(1, max(2, n // 2)))
    where = rng.choice(n, size=m, replace=False)
    if kind == "duplicates":  # copies of a few rows, scattered
        src = rng.choice(n, size=max(1, m // 8), replace=False)
        rows[where] = rows[rng.choice(src, size=m)]
    elif kind == "cluster":  # many rows within 1e-6 .. 1e-3 of the query direction
        scale = 10.0 ** rng.uniform(-6, -3)
        for i in where:
            rows[i] = _unit(q0.astype(np.float64) + scale * rng.standard_normal(384))
    elif kind == "sparse":
        for i in where:
            v = np.zeros(384)
            idx = rng.choice(384, size=int(rng.integers(1, 5)), replace=False)
            v[idx] = rng.standard_normal(len(idx))
            v += 10.0 ** rng.uniform(-9, -5) * rng.standard_normal(384)
            rows[i] = _unit(v)
    elif kind == "onehot":  # a sub-tile holding one of these is quantised with scale 1/127
        for i in where:
            v = np.zeros(384)
            idx = rng.choice(384, size=int(rng.integers(1, 3)), replace=False)
            v[idx] = rng.choice([-1.0, 1.0], size=len(idx)) * rng.uniform(0.3, 1.0, size=len(idx))
            rows[i] = _unit(v)
        rows[where[: max(1, m // 3)]] = np.stack([_unit(rng.standard_normal(384)) for _ in range(max(1, m // 3))])
    elif kind == "heavy":  # bell-shaped rows with a few dimensions far larger than the rest (sentence embeddings); the
        # whole index, not a subset: top scores then crowd within the int8 bound's slack (deepening rounds)
        g = rng.standard_normal((n, 384))
        dims = rng.choice(384, size=int(rng.integers(1, 6)), replace=False)
        g[:, dims] *= rng.uniform(3, 20)
        rows = np.stack([_unit(v) for v in g])
    elif kind == "antipodal":
        rows[where] = -rows[rng.choice(n, size=m)]
        rows[where[: max(1, m // 4)]] = -q0
    return rows


import os  # noqa: E402

_EXAMPLES = int(os.environ.get("DAWN_HYP_EXAMPLES", "150"))  # soak runs: DAWN_HYP_EXAMPLES=3000


@settings(max_examples=_EXAMPLES, deadline=None,
          suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])
@given(seed=st.integers(0, 2**31 - 1), n=st.one_of(st.integers(1, 2500), st.integers(8000, 30000)), k=st.integers(1, 64),
       B=st.sampled_from([1, 2, 3, 4, 7, 9, 33, 70]),
       kind=st.sampled_from(["random", "duplicates", "cluster", "sparse", "antipodal", "onehot", "heavy"]),
       dtype=st.sampled_from(["f32", "bf16"]), force_stream=st.booleans(), sched=st.sampled_from([4, 5, 1]),
       i8=st.booleans(), shards=st.sampled_from([0, 0, 2, 3]), packed=st.sampled_from([0, 0, 5, 6]))
def test_any_index_any_batch_matches_the_oracle(dawn, oracle, seed, n, k, B, kind, dtype, force_stream, sched, i8, shards, packed):
    rng = np.random.default_rng(seed)
    Q = synth.unit_rows(int(rng.integers(1, 1 << 30)), 0, B)
    rows = _build_rows(rng, n, kind, Q[0])
    if kind == "heavy":  # queries of the same kind
        Q = _build_rows(rng, B, "heavy", Q[0]) if B >= 4 else Q
    if kind in ("duplicates", "cluster", "onehot") and n >= 2:
        Q[B - 1] = rows[int(rng.integers(0, n))]  # a query that IS a row
    ids = rng.permutation(np.arange(10, 10 + n)).astype(np.uint64)  # labels are arbitrary, order of insertion rules ties
    # shards > 0: the same index behind one sharded handle (logical shards on this device, small chunks)
    idx = dawn.VectorIndex(0, dtype=dtype) if not shards else dawn.VectorIndex(devices=[0] * shards, dtype=dtype)
    try:
        if shards:
            idx.set_option("shard_chunk", 64 * int(rng.integers(1, 9)))
        idx.add_batch(ids, rows)
        stored = synth.round_bf16(rows) if dtype == "bf16" else rows
        idx.set_option("i8_shadow", int(i8))  # False: filter on the f16 shadow / the bf16 rows themselves
        if packed:  # single queries stream the packed shadow (scan_i6.hip) whatever the size of the index
            idx.set_option("i6_bits", packed)
            idx.set_option("i6_min_rows", 0)
        if force_stream:
            idx.set_option("mfma_min_batch", 100000)
        idx.set_option("mfma_sched", sched)  # 5: the pipelined matrix-core kernel for every pass, 1: the 8-wave kernel
        labels, dist, found = idx.search_batch(Q, k)
        if packed and B > 1:  # the batch went to the matrix-core pass / the int8 stream: every query once more on its own
            for b in range(B):
                l1, d1 = idx.search(Q[b], k)
                assert np.array_equal(l1, labels[b][:found[b]]) and np.array_equal(d1.view(np.uint32), dist[b][:found[b]].view(np.uint32)), \
                    (kind, dtype, n, k, B, b, packed)
        for b in range(B):
            olab, odist = oracle.scan_topk(stored, ids, Q[b], k)
            assert found[b] == min(k, n)
            assert np.array_equal(labels[b][:found[b]], olab), (kind, dtype, n, k, B, b)
            assert np.array_equal(dist[b][:found[b]].view(np.uint32), odist.view(np.uint32)), (kind, dtype, n, k, B, b)
    finally:
        idx.close()
