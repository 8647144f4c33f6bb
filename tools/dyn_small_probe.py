"""Static interleave against the dynamically assigned tail of the packed stream on small indexes (dev tool).  python tools/dyn_small_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, "/root/repo")
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream().cuda_stream
NQ = 256; Q = synth.unit_rows(11, 0, NQ); d_q = torch.from_numpy(Q).to(dev); k = 10
blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev); p = blob.data_ptr()
for rows in (1_000_000, 1_500_000, 2_000_000, 3_000_000, 4_000_000, 6_000_000, 12_500_000):
    idx = dawn.VectorIndex(0); idx.fill_synthetic(1, 0, rows, 1)
    def timed(label):
        for i in range(8): idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for rep in range(4):
            for i in range(NQ): idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
        torch.cuda.synchronize()
        print(f"rows={rows} {label}: {(time.perf_counter() - t0) / (4 * NQ) * 1e3:.4f} ms", flush=True)
    idx.set_option("stream_dynamic_tail", 0); timed("static only          ")
    idx.set_option("stream_dynamic_tail", 1); timed("default (chunk 8, 3/16)")
    idx.close()
