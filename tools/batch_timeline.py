"""One topical batch of 256 under `rocprofv3 --kernel-trace` (dev tool): warm batches until the feedback has settled, a pause, then
`n` batches — tools/profile_summary.py timeline <db> <out> <window_ms> prints the kernels of the last ones.
python tools/batch_timeline.py [rows=100000000] [dist=4] [k=10] [n=2] [B=256]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2
B = int(sys.argv[5]) if len(sys.argv) > 5 else 256
dev = torch.device("cuda", 0)
qi = dawn.VectorIndex(0)
qi.set_option("synth_dist", dist)
qi.fill_synthetic(1, 1 << 40, 256 * 256, 1)
Q = qi.get_rows(0, 256 * 256)[0][::256].copy()[:B]
qi.close()
idx = dawn.VectorIndex(0)
idx.set_option("synth_dist", dist)
idx.fill_synthetic(1, 0, rows, 1)
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
blob = torch.zeros((dawn.result_blob_bytes(B, k),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
for _ in range(8):
    idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
torch.cuda.synchronize()
time.sleep(1.0)
r0 = idx.stats_raw()
t0 = time.perf_counter()
for _ in range(n):
    idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / n * 1e3
r1 = idx.stats_raw()
print(f"rows={rows} dist={dist} k={k} B={B}: {el:.2f} ms per batch; per batch: {[(b - a) / n for a, b in zip(r0, r1)]}; feedback {idx.stats_batch_feedback()}")
