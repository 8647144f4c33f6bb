"""Randomised cross-check of the search paths this round added or touched (GPU): random index sizes (around every tile boundary the
kernels have: 16, 32, 128 rows, groups of 8 tiles), row kinds, k, batch sizes and option sets — FP6 first filter and its row /
int8 re-scoring, the bounded pass on the packed shadow with and without its seed, the batch's second pass, forced ladders,
demotion, deepened thresholds — every answer compared bit for bit with the CPU oracle's scan of the same rows
(oracle/dawn_oracle.c: src/search/vector.rs:128-134 + exact top-k, ties to the lower position).  Seeds are fixed: a failure
reproduces."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from dawnsearch_amd import synth  # noqa: E402

from test_scan_gpu import _assert_same  # noqa: E402

QROW0 = 1 << 40


def _case(rng):
    edges = [1, 15, 16, 17, 31, 32, 33, 127, 128, 129, 1023, 1024, 1025, 8191, 8192, 8193]
    if rng.random() < 0.35:
        n = int(rng.choice(edges)) * int(rng.choice([1, 1, 3, 8, 16]))
    else:
        n = int(rng.integers(1, 260_000))
    n = max(1, min(n, 260_000))
    dist = int(rng.choice([0, 0, 4, 5]))
    dtype = "bf16" if (dist == 0 and rng.random() < 0.25) else "f32"
    k = int(rng.choice([1, 5, 10, 20, 33, 64]))
    B = int(rng.choice([1, 1, 3, 17, 64, 256, 300, 513]))  # (> 256: several passes behind one call)
    if n > 120_000 and B > 64:
        B = 64  # (the oracle's side of the comparison: n x B x 384 on the host cores)
    if n > 40_000 and B > 256:
        B = 256
    opts = {}
    if rng.random() < 0.5:
        opts["i6_min_rows"] = 0
    if rng.random() < 0.5:
        opts["f6_min_rows"] = 0
        opts["f6_shadow"] = 1
        if rng.random() < 0.4:
            opts["f6_refine_rows"] = 0
        if rng.random() < 0.3:
            opts["f6_target"] = int(rng.choice([256, 1024, 24576]))
        if rng.random() < 0.2:
            opts["f6_stagger"] = int(rng.choice([0, 8]))
    if rng.random() < 0.5:
        opts["bounded_packed"] = int(rng.choice([0, 2]))
    if rng.random() < 0.5:
        opts["bounded_seed"] = int(rng.choice([0, 2]))
        if rng.random() < 0.4:
            opts["bounded_seed_shift"] = int(rng.choice([2, 4, 7, 8]))
    if rng.random() < 0.4:
        opts["batch_rerun"] = 2
    if rng.random() < 0.35:
        opts["force_fallback"] = 2
    if rng.random() < 0.4:
        opts["ladder_feedback"] = int(rng.choice([0, 2]))
    if rng.random() < 0.4:
        opts["mfma_target"] = int(rng.choice([64, 256, 4096]))
    if rng.random() < 0.3:
        opts["bounded_multi_waves"] = int(rng.choice([4, 8]))
    if rng.random() < 0.3:
        opts["bounded_multi_packed"] = 1
    return n, dist, dtype, k, B, opts


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_configuration_with_adds_and_shards(dawn, oracle, seed):
    """The same on indexes that GROW between searches (rows appended in two batches: every shadow re-quantises its last partial
    tile) and on sharded handles (three logical shards on one device, rows dealt in chunks): uniform rows, f32."""
    rng = np.random.default_rng(5000 + seed)
    n, _, _, k, B, opts = _case(rng)
    n = max(n, 200)
    B = min(B, 64)
    sharded = rng.random() < 0.5
    idx = dawn.VectorIndex(devices=[0, 0, 0]) if sharded else dawn.VectorIndex(0)
    if sharded:
        idx.set_option("shard_chunk", int(rng.choice([64, 1024, 4096])))
        opts.pop("bounded_multi_waves", None)
        opts.pop("bounded_multi_packed", None)
    for name, v in opts.items():
        idx.set_option(name, v)
    x = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    cuts = sorted({int(n * rng.uniform(0.3, 0.8)), int(n * rng.uniform(0.8, 0.999)), n})
    Q = synth.unit_rows(9 + seed, 0, B)
    Q[0] = synth.planted_queries(1, [n - 1], 7)[0]
    try:
        lo = 0
        for hi in cuts:
            if hi <= lo:
                continue
            idx.add_batch(ids[lo:hi], x[lo:hi])
            lo = hi
            if B == 1:
                lab, dd = idx.search(Q[0], k)
                _assert_same(lab, dd, *oracle.scan_topk(x[:hi], ids[:hi], Q[0], k, threads=8))
            else:
                lab, dd, found = idx.search_batch(Q, k)
                for b in range(B):
                    assert found[b] == min(k, hi)
                    _assert_same(lab[b][:found[b]], dd[b][:found[b]], *oracle.scan_topk(x[:hi], ids[:hi], Q[b], k, threads=8))
        assert idx.size() == n
    finally:
        idx.close()


@pytest.mark.parametrize("seed", list(range(160)))
def test_random_configuration_equals_the_oracle(dawn, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n, dist, dtype, k, B, opts = _case(rng)
    idx = dawn.VectorIndex(0, dtype=dtype)
    if dist:
        idx.set_option("synth_dist", dist)
    for name, v in opts.items():
        idx.set_option(name, v)
    idx.fill_synthetic(1, 0, n, 1)
    if dist:
        Q = np.concatenate([synth.unit_rows_topical(1, QROW0 + 256 * i, 1, runs=(dist == 5)) for i in range(B)])
        want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=dist)
        wl = [want[0][b][:min(k, n)] for b in range(B)]
        wd = [want[1][b][:min(k, n)] for b in range(B)]
    else:
        x = oracle.unit_rows(1, 0, n)
        if dtype == "bf16":
            x = synth.round_bf16(x)
        ids = np.arange(1, n + 1, dtype=np.uint64)
        Q = synth.unit_rows(2 + seed, 0, B)
        Q[0] = synth.planted_queries(1, [int(rng.integers(0, n))], 7)[0]
        res = [oracle.scan_topk(x, ids, q, k, threads=8) for q in Q]
        wl = [r[0] for r in res]
        wd = [r[1] for r in res]
    try:
        for rep in range(2):  # (twice: the second call runs with whatever the feedback learned from the first)
            if B == 1:
                lab, dd = idx.search(Q[0], k)
                _assert_same(lab, dd, wl[0], wd[0])
            else:
                lab, dd, found = idx.search_batch(Q, k)
                for b in range(B):
                    assert found[b] == min(k, n), (n, dist, dtype, k, B, opts, b)
                    _assert_same(lab[b][:found[b]], dd[b][:found[b]], wl[b], wd[b])
        st = idx.stats()
        assert st["fallbacks"] == 0 or dtype == "bf16" or "force_fallback" in opts or n <= 64, (st, n, dist, dtype, k, B, opts)
    finally:
        idx.set_option("bounded_multi_waves", 8)  # (process-wide knobs: back to the defaults for the tests that follow)
        idx.set_option("bounded_multi_packed", 0)
        idx.close()


_TOGGLES = {
    "i8_shadow": [0, 1], "i6_shadow": [0, 1], "i6_bits": [5, 6], "f16_shadow": [0, 1], "f16_shadow_b1": [0, 1], "f6_shadow": [0, 1],
    "i8_batched": [0, 1], "mfma_min_batch": [0, 2, 100000], "bounded_pass": [0, 1], "force_fallback": [0, 2], "ladder_feedback": [0, 1, 2],
    "bounded_packed": [0, 1, 2], "bounded_seed": [0, 1, 2], "bounded_seed_shift": [2, 5, 8], "batch_rerun": [0, 1, 2], "mfma_target": [64, 1024, 4096],
    "i6_slack_model": [0, 1], "zero_copy_batch": [0, 8, 256], "i6_central_tail": [0, 1], "stream_dynamic_tail": [0, 1], "bounded_multi_packed": [0, 1], "i6_refine": [0, 8, 64], "f6_refine_rows": [0, 1], "f6_target": [256, 12288],
}


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_option_walk_on_one_index(dawn, oracle, seed, tmp_path):
    """One index, a random walk through its options between searches (shadows are built and released, kernels change, ladders are
    forced and released): after every step a single query and a batch equal the oracle."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([5000, 33_333, 70_000, 150_001]))
    dist = int(rng.choice([0, 4]))
    k = int(rng.choice([10, 20, 64]))
    idx = dawn.VectorIndex(0)
    if dist:
        idx.set_option("synth_dist", dist)
    idx.set_option("i6_min_rows", 0)
    idx.set_option("f6_min_rows", 0)
    idx.fill_synthetic(1, 0, n, 1)
    B = 24
    if dist:
        Q = np.concatenate([synth.unit_rows_topical(1, QROW0 + 256 * i, 1) for i in range(B)])
        want = oracle.scan_topk_synth(1, 0, n, 1, Q, k, dist=dist)
        wl, wd = list(want[0]), list(want[1])
    else:
        x = oracle.unit_rows(1, 0, n)
        ids = np.arange(1, n + 1, dtype=np.uint64)
        Q = synth.unit_rows(3 + seed, 0, B)
        res = [oracle.scan_topk(x, ids, q, k, threads=8) for q in Q]
        wl, wd = [r[0] for r in res], [r[1] for r in res]
    names = sorted(_TOGGLES)
    trail = []
    try:
        for step in range(10):
            for _ in range(int(rng.integers(1, 4))):
                name = names[int(rng.integers(0, len(names)))]
                v = int(rng.choice(_TOGGLES[name]))
                idx.set_option(name, v)
                trail.append((name, v))
            b = int(rng.integers(0, B))
            lab, dd = idx.search(Q[b], k)
            assert np.array_equal(lab, wl[b]) and np.array_equal(dd.view(np.uint32), np.asarray(wd[b]).view(np.uint32)), (n, dist, k, trail)
            labs, dds, found = idx.search_batch(Q, k)
            for j in range(B):
                assert found[j] == k
                assert np.array_equal(labs[j], wl[j]) and np.array_equal(dds[j].view(np.uint32), np.asarray(wd[j]).view(np.uint32)), (n, dist, k, j, trail)
            if step % 3 == 1:
                # a peer's search: only the hits with distance < limit (udp_service.rs:196-199)
                limit = float(np.asarray(wd[b])[int(rng.integers(0, k))])
                keep = int(np.sum(np.asarray(wd[b]) < np.float32(limit)))
                lab, dd = idx.search_limited(Q[b], k, limit)
                assert len(lab) == keep and np.array_equal(lab, wl[b][:keep]), (n, dist, k, b, limit, trail)
            if step == 6:
                # save -> load into a fresh index (its shadows are rebuilt from the rows): the same answers
                path = os.path.join(tmp_path, "walk.dawn")
                idx.save(path)
                other = dawn.VectorIndex(0)
                try:
                    other.load(path)
                    assert other.size() == n
                    lab, dd = other.search(Q[b], k)
                    assert np.array_equal(lab, wl[b]) and np.array_equal(dd.view(np.uint32), np.asarray(wd[b]).view(np.uint32)), (n, dist, k, trail)
                finally:
                    other.close()
    finally:
        idx.set_option("bounded_multi_packed", 0)
        idx.close()
