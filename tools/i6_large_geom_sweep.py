"""Loads in flight and the dynamic tail's chunks of the packed stream on large indexes (dev tool): ms per single-query search.
python tools/i6_large_geom_sweep.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [25_000_000, 50_000_000, 100_000_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
NQ = 64
Q = synth.unit_rows(11, 0, NQ)
d_q = torch.from_numpy(Q).to(dev)
k = 10
blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
for rows in sizes:
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, rows, 1)
    idx.set_option("ladder_feedback", 0)

    def timed(label):
        for i in range(6):
            idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(2):
            for i in range(NQ):
                idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / (2 * NQ) * 1e3
        print(f"rows={rows} {label}: {el:.4f} ms per search", flush=True)

    timed("default geometry     ")
    for ring in (4, 8):
        for chunk, share in ((8, 2), (8, 3), (16, 1), (16, 2), (16, 3), (32, 2), (32, 3)):
            idx.set_option("i6_scan_threads", 512)
            idx.set_option("i6_scan_ring", ring)
            idx.set_option("i6_dyn_chunk", chunk)
            idx.set_option("i6_dyn_share", share)
            timed(f"ring={ring} chunk={chunk:2d} share={share}/16")
    idx.close()
    torch.cuda.synchronize()
    time.sleep(1.5)
