"""Certificate operating points against the ladder's real cost (round-4 verdict item 3): ms per device-resident batch of 256 (and per single query)
vs "mfma_target" on uniform and topical rows, with the share of queries the ladder answered.  python tools/operating_point_sweep.py rows [dists=0,4]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dists = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4]
targets = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [128, 256, 512, 1024, 2048, 4096]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for dist in dists:
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    if dist >= 4:
        qi.fill_synthetic(1, 1 << 40, 256 * 256, 1)
        Q = qi.get_rows(0, 256 * 256)[0][::256].copy()
    else:
        qi.fill_synthetic(2, 0, 256, 1)
        Q = qi.get_rows(0, 256)[0]
    qi.close()
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", dist)
    idx.fill_synthetic(1, 0, rows, 1)
    idx.set_option("ladder_feedback", 0)  # (the depth is set by hand)
    d_q = torch.from_numpy(Q).to(dev)
    for k in (10, 20):
        blob = torch.zeros((dawn.result_blob_bytes(256, k),), dtype=torch.uint8, device=dev)
        p = blob.data_ptr()
        ref = None
        for target in targets:
            idx.set_option("mfma_target", target)
            for _ in range(2):
                idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
            torch.cuda.synchronize()
            r0 = idx.stats_raw()
            n = 8
            t0 = time.perf_counter()
            for _ in range(n):
                idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / n * 1e3
            r1 = idx.stats_raw()
            out = blob.cpu().numpy()[:256 * k * 12].copy()
            ref = out if ref is None else ref
            d = [(b - a) / n for a, b in zip(r0, r1)]
            print(f"rows={rows} dist={dist} k={k} target={target:5d}: {el:7.3f} ms per batch; second chance {d[2]:5.1f} deep {d[3]:5.1f} bounded {d[4]:5.1f} "
                  f"fallbacks {d[1]:.0f}; same answers {bool(np.array_equal(out, ref))}", flush=True)
    idx.close()
