"""A/B of the packed stream's dynamic tail (dev tool): chunk size x dynamic share, interleaved rounds, device-resident searches between
two synchronisations + the stream kernel's own duration.  python tools/stream_chunk_ab.py [rows=12500000] [rounds=3]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
idx = dawn.VectorIndex(0)
idx.set_option("i6_min_rows", 0)
idx.fill_synthetic(1, 0, rows, 1)
Q = np.concatenate([synth.unit_rows(2, 0, 7), synth.planted_queries(1, [rows // 3], 4)])
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
iters = 30 if rows > 30_000_000 else 200
cfgs = [(16, 2), (8, 2), (4, 2), (2, 2), (8, 3), (4, 3), (4, 4), (2, 4), (1, 2)]
want = [idx.search(q, 10) for q in Q]
res = {c: [] for c in cfgs}
ker = {c: [] for c in cfgs}
blob = torch.zeros((dawn.result_blob_bytes(1, 10),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
for r in range(rounds):
    for c in cfgs:
        idx.set_option("i6_dyn_chunk", c[0])
        idx.set_option("i6_dyn_share", c[1])
        if r == 0:
            got = [idx.search(q, 10) for q in Q]
            assert all(np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32)) for g, w in zip(got, want)), c
        for i in range(5):
            idx.search_device(d_q.data_ptr() + (i % 8) * 1536, 1, 10, p, p + 80, p + 120, stream)
        idx.profile_enable(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            idx.search_device(d_q.data_ptr() + (i % 8) * 1536, 1, 10, p, p + 80, p + 120, stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        n, ms = idx.profile_read()
        idx.profile_enable(False)
        res[c].append(el / iters * 1e3)
        ker[c].append(ms / max(n, 1))
for c in cfgs:
    print(f"rows={rows} chunk {c[0]:3d} share {c[1]:2d}/16: ms per search {[round(v, 4) for v in res[c]]} best {min(res[c]):.4f}; "
          f"stream kernel us {[round(v * 1e3, 1) for v in ker[c]]}", flush=True)
print(idx.stats())
