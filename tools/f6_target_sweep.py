"""FP6 first filter: ms per batch of 256 vs "f6_target" (survivors per query its threshold aims for), uniform rows (dev tool).
python tools/f6_target_sweep.py [rows=100000000]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
idx = dawn.VectorIndex(0)
idx.set_option("f6_min_rows", 0)
idx.set_option("f6_shadow", 1)
idx.fill_synthetic(1, 0, rows, 1)
idx.set_option("ladder_feedback", 0)
Q = synth.unit_rows(3, 0, 256)
d_q = torch.from_numpy(Q).to(dev)
for k in (10, 20):
    blob = torch.zeros((dawn.result_blob_bytes(256, k),), dtype=torch.uint8, device=dev)
    p = blob.data_ptr()
    ref = None
    for target in (12288, 8192, 6144, 4096, 3072, 2048, 1024):
        idx.set_option("f6_target", target)
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
        print(f"rows={rows} k={k} f6_target={target:6d}: {el:7.3f} ms per batch; second chance {d[2]:5.1f} deep {d[3]:5.1f} bounded {d[4]:5.1f}; "
              f"same answers {bool(np.array_equal(out, ref))}", flush=True)
