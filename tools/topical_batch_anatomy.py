"""Where a topical batch of 256 spends its time (dev tool): 100 M rows of synth_dist 4, queries = further rows of the same stream; per
batch: ms, queries the bounded pass answered, (row, query) pairs it scored exactly (dawn_index_debug_raw_stats[7]); with the batch
feedback's deepened thresholds and without.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
python tools/topical_batch_anatomy.py [rows=100000000] [dist=4] [batches=4]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
opts = [a.split("=") for a in sys.argv[4:]]
dev = torch.device("cuda", 0)
qi = dawn.VectorIndex(0)
qi.set_option("synth_dist", dist)
qi.fill_synthetic(1, 1 << 40, 256 * 256, 1)
Q = qi.get_rows(0, 256 * 256)[0][::256].copy()
qi.close()
idx = dawn.VectorIndex(0)
idx.set_option("synth_dist", dist)
idx.fill_synthetic(1, 0, rows, 1)
for n, v in opts:
    idx.set_option(n, int(v))
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
for k in (10, 20):
    blob = torch.zeros((dawn.result_blob_bytes(256, k),), dtype=torch.uint8, device=dev)
    p = blob.data_ptr()
    for target in (1024, 4096):
        idx.set_option("mfma_target", target)
        idx.set_option("ladder_feedback", 0)  # (the depth is set by hand here)
        idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
        torch.cuda.synchronize()
        for b in range(batches):
            r0 = idx.stats_raw()
            t0 = time.perf_counter()
            idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) * 1e3
            r1 = idx.stats_raw()
            d = [b1 - a1 for a1, b1 in zip(r0, r1)]
            print(f"rows={rows} dist={dist} k={k} target={target}: {el:8.2f} ms; bounded {d[4]:3d} (streams of 16: {-(-d[4] // 16)}), second chance "
                  f"{d[2]}, deep {d[3]}, exact pairs {d[7] / 1e6:.2f} M, fallbacks {d[1]}", flush=True)
