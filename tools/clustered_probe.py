"""Topical-mixture probe (dev tool): an index of synth_dist 4 / 5 rows (dawnsearch_amd/synth.py: unit_rows_topical), queries =
FURTHER rows of the same stream (new pages on the same topics: row numbers 2^40 + i), k = 10 / 20, batch 1 and 256: ms per search,
which rung of the ladder answered (dawn_index_stats*), every batch-1 answer compared bit for bit with the exact pass's
(force_fallback) on a sample.  python tools/clustered_probe.py [rows=12500000] [queries=64] [dists=4,5]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dists = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4, 5]
QROW0 = 1 << 40

for dist in dists:
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", dist)
    t0 = time.perf_counter()
    idx.fill_synthetic(1, 0, rows, 1)
    fill_s = time.perf_counter() - t0
    # the generator on the GPU = the numpy / C restatement, bit for bit
    got = idx.get_rows(rows - 300, 300)[0]
    want = synth.unit_rows_topical(1, rows - 300, 300, runs=(dist == 5))
    gen_ok = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    nqq = max(nq, 256)
    qi.fill_synthetic(1, QROW0, nqq * 256, 1)  # (every 256th row: a different run — site — each, for dist 5)
    Q = qi.get_rows(0, nqq * 256)[0][::256].copy()
    qi.close()
    cl, tl = synth.topical_cluster(1, QROW0 + 256 * np.arange(nq), runs=(dist == 5))
    print(f"dist={dist} rows={rows}: fill {fill_s:.1f} s, generator == numpy: {gen_ok}; query clusters (first 16) "
          f"{cl[:16].tolist()}", flush=True)
    for mode, bounded, i6 in (("packed stream, exact pass behind it (round 3)", 0, 1), ("packed stream + bounded pass", 1, 1),
                              ("int8 stream, exact pass behind it", 0, 0), ("int8 stream + bounded pass", 1, 0)):
        idx.set_option("bounded_pass", bounded)
        idx.set_option("i6_shadow", i6)
        print(f" {mode}:", flush=True)
        for k in (10, 20):
            s0 = idx.stats()
            per = []
            for q in Q[:nq]:
                t0 = time.perf_counter()
                idx.search(q, k)
                per.append(time.perf_counter() - t0)
            s1 = idx.stats()
            per = np.array(per) * 1e3
            print(f"  k={k:2d} batch 1: {nq} queries, ms per search p50 {np.percentile(per, 50):.3f} mean {per.mean():.3f} "
                  f"max {per.max():.3f}; { {kk: s1[kk] - s0[kk] for kk in s1} }", flush=True)
            if i6 == 1:  # (batches never read the packed shadow)
                continue
            s0 = idx.stats()
            idx.search_batch(Q[:256], k)
            t0 = time.perf_counter()
            idx.search_batch(Q[:256], k)
            el = time.perf_counter() - t0
            s1 = idx.stats()
            print(f"  k={k:2d} batch 256: {el * 1e3:.2f} ms; two batches: { {kk: s1[kk] - s0[kk] for kk in s1} }", flush=True)
    idx.set_option("bounded_pass", 1)
    idx.set_option("i6_shadow", 1)
    # parity of a sample with the exact pass
    n_chk = min(nq, 8)
    got = [idx.search(q, 10) for q in Q[:n_chk]]
    idx.set_option("force_fallback", 1)
    want = [idx.search(q, 10) for q in Q[:n_chk]]
    idx.set_option("force_fallback", 0)
    same = all(np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) for g, w in zip(got, want))
    print(f"  {n_chk} answers identical to the exact pass: {same}; top-10 distances of query 0: {np.round(got[0][1], 4).tolist()}",
          flush=True)
    idx.close()
