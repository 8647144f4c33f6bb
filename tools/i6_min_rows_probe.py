"""Where the packed stream starts to pay (dev tool): ms per single-query search with the packed shadow ("i6_min_rows" = 0) and with
the int8 stream, per index size.   python tools/i6_min_rows_probe.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [100_000, 250_000, 500_000, 1_000_000, 1_500_000, 2_000_000, 3_000_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
NQ = 256
Q = synth.unit_rows(11, 0, NQ)
d_q = torch.from_numpy(Q).to(dev)
for rows in sizes:
    for k in (10, 20):
        res = {}
        for packed in (1, 0):
            idx = dawn.VectorIndex(0)
            idx.set_option("i6_min_rows", 0 if packed else 1 << 40)
            idx.fill_synthetic(1, 0, rows, 1)
            blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
            p = blob.data_ptr()
            for i in range(8):
                idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
            torch.cuda.synchronize()
            r0 = idx.stats_raw()
            t0 = time.perf_counter()
            for rep in range(4):
                for i in range(NQ):
                    idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / (4 * NQ) * 1e3
            r1 = idx.stats_raw()
            res[packed] = (el, [b - a for a, b in zip(r0, r1)], idx.i6_refine(k)[0])
            idx.close()
        print(f"rows={rows} k={k}: packed {res[1][0]:.4f} ms (lists of {res[1][2]}, ladder {res[1][1][1:6]}), int8 stream {res[0][0]:.4f} ms "
              f"(ladder {res[0][1][1:6]}): packed / int8 = {res[1][0] / res[0][0]:.3f}", flush=True)
