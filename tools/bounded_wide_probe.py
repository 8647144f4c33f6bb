"""The wide batch form of the bounded pass alone (dev tool): uniform rows (almost nothing passes the int8 bound: the pass is its stream +
its MFMAs) and topical rows (rows queued, re-tested on f32, appended), B flagged queries forced through the ladder ("force_fallback" = 2).
Run under rocprofv3 --kernel-trace --stats for the kernel's own time.  python tools/bounded_wide_probe.py [rows=100000000] [dists=0,4]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dists = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 4]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
k = 10
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
    idx.set_option("force_fallback", 2)
    idx.set_option("ladder_feedback", 0)
    d_q = torch.from_numpy(Q).to(dev)
    for B in (32, 64, 128, 256):
        blob = torch.zeros((dawn.result_blob_bytes(B, k),), dtype=torch.uint8, device=dev)
        p = blob.data_ptr()
        for wide in (1, 0) if B <= 64 else (1,):
            idx.set_option("bounded_wide", wide)
            idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()
            r0 = idx.stats_raw()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / n * 1e3
            r1 = idx.stats_raw()
            print(f"rows={rows} dist={dist} B={B} wide={wide}: {el:8.2f} ms per batch (first pass + tail + ladder); wide answers "
                  f"{(r1[0] - r0[0]) // n}, bounded {(r1[4] - r0[4]) // n}, pairs {(r1[7] - r0[7]) / n / 1e6:.2f} M", flush=True)
    idx.close()
