"""Does the headline stream (packed 5-bit shadow, one query) run faster once the device has settled after the fill's frees?  (dev tool)
python tools/headline_settle_probe.py [rows=100000000]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
q = torch.from_numpy(synth.unit_rows(2, 0, 1)).to(dev)
k = 10
blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
t0 = time.time()


def leg(tag, steps=20, warm=3):
    idx.profile_enable(True)
    for _ in range(warm):
        idx.search_device(q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)
    torch.cuda.synchronize()
    idx.profile_read()
    for _ in range(steps):
        idx.search_device(q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)
    torch.cuda.synchronize()
    n, ms = idx.profile_read()
    idx.profile_enable(False)
    print(f"{time.time() - t0:6.1f} s {tag:50s}: packed stream {ms / n:7.4f} ms = {rows * 240.25 / (ms / n * 1e-3) / 8e12:.4f} of HBM", flush=True)


leg("right after the fill")
leg("again")
time.sleep(3)
leg("after 3 s of idle")
leg("again")
for B in (256,):
    Q = torch.from_numpy(synth.unit_rows(3, 0, B)).to(dev)
    bl = torch.zeros((dawn.result_blob_bytes(B, k),), dtype=torch.uint8, device=dev)
    pp = bl.data_ptr()
    for _ in range(5):
        idx.search_device(Q.data_ptr(), B, k, pp, pp + B * k * 8, pp + B * k * 12, stream)
    torch.cuda.synchronize()
leg("after five batches of 256")
time.sleep(3)
leg("after 3 s of idle")
