"""A/B of the packed stream's tail (dev tool): every workgroup rescoring its own 64 rows exactly in the stream's epilogue +
merge_exact_kernel (round 3) against the central tail (option i6_central_tail = 1: the refined lists go to merge_rescore_kernel,
which rescores the index's 64 best rows).  Interleaved rounds, device-resident searches timed between two synchronisations, the
stream kernel's own duration from its HIP events, results compared bit for bit.
python tools/stream_central_tail_ab.py [rows=12500000] [rounds=3]"""
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
res = {0: [], 1: []}
ker = {0: [], 1: []}
for k in (10, 20):
    want = None
    for mode in (0, 1):
        idx.set_option("i6_central_tail", mode)
        got = [idx.search(q, k) for q in Q]
        if want is None:
            want = got
        else:
            assert all(np.array_equal(g[0], w[0]) and np.array_equal(g[1].view(np.uint32), w[1].view(np.uint32))
                       for g, w in zip(got, want)), (k, mode)
blob = torch.zeros((dawn.result_blob_bytes(1, 10),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
for r in range(rounds):
    for mode in (0, 1):
        idx.set_option("i6_central_tail", mode)
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
        res[mode].append(el / iters * 1e3)
        ker[mode].append(ms / max(n, 1))
for mode, name in ((0, "exact rescore in the stream's epilogue + merge_exact_kernel"), (1, "central tail (merge_rescore_kernel)")):
    print(f"rows={rows} {name:62s}: ms per search {[round(v, 4) for v in res[mode]]} best {min(res[mode]):.4f}; "
          f"stream kernel us {[round(v * 1e3, 1) for v in ker[mode]]}", flush=True)
print(idx.stats())
