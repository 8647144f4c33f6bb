"""FP6 first filter on shard-sized indexes (12.5 M / 25 M / 50 M rows): ms per batch of 256 with it and with the int8 first
filter, over "f6_target" (dev tool; decides "f6_min_rows").   python tools/f6_small_probe.py [rows ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [12_500_000, 25_000_000, 50_000_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
Q = synth.unit_rows(3, 0, 256)
d_q = torch.from_numpy(Q).to(dev)


def timed(idx, k, blob, n=12):
    p = blob.data_ptr()
    for _ in range(3):
        idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
    torch.cuda.synchronize()
    r0 = idx.stats_raw()
    t0 = time.perf_counter()
    for _ in range(n):
        idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n * 1e3
    r1 = idx.stats_raw()
    return el, [(b - a) / n for a, b in zip(r0, r1)], blob.cpu().numpy()[:256 * k * 12].copy()


for rows in sizes:
    idx = dawn.VectorIndex(0)
    idx.set_option("f6_min_rows", 0)
    idx.set_option("f6_shadow", 1)
    idx.fill_synthetic(1, 0, rows, 1)
    idx.set_option("ladder_feedback", 0)
    for k in (10, 20):
        blob = torch.zeros((dawn.result_blob_bytes(256, k),), dtype=torch.uint8, device=dev)
        idx.set_option("f6_shadow", 0)
        torch.cuda.synchronize()
        time.sleep(2.0)
        el, d, ref = timed(idx, k, blob)
        print(f"rows={rows} k={k} int8 first filter      : {el:7.3f} ms per batch; second chance {d[2]:5.1f} deep {d[3]:5.1f} bounded {d[4]:5.1f}", flush=True)
        idx.set_option("f6_shadow", 1)
        for target in (12288, 6144, 3072, 1536):
            idx.set_option("f6_target", target)
            el, d, out = timed(idx, k, blob)
            fb = idx.stats_batch_feedback()
            print(f"rows={rows} k={k} fp6 f6_target={target:6d}: {el:7.3f} ms per batch; second chance {d[2]:5.1f} deep {d[3]:5.1f} bounded {d[4]:5.1f}; "
                  f"f6_batches {fb['f6_batches']}; same answers {bool(np.array_equal(out, ref))}", flush=True)
    idx.close()
    torch.cuda.synchronize()
    time.sleep(2.0)
