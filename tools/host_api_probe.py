"""The synchronous host API (dawn_index_search_batch: host buffers in and out, what the reference's index.search is) against the
device-resident call, per batch size; results by zero-copy stores or by copy commands (option "zero_copy_batch") (dev tool).
python tools/host_api_probe.py [rows ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 12_500_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
k = 10
for rows in sizes:
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, rows, 1)
    for B in (1, 8, 9, 64, 256):
        Q = synth.unit_rows(11, 0, B)
        d_q = torch.from_numpy(Q).to(dev)
        blob = torch.zeros((dawn.result_blob_bytes(B, k),), dtype=torch.uint8, device=dev)
        p = blob.data_ptr()

        def p50(fn, calls=200):
            ts = []
            for i in range(calls + 20):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            return float(np.percentile(np.array(ts[20:]) * 1e3, 50))

        def dev_call():
            idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()

        t_dev = p50(dev_call)
        res = {}
        for zc in (256, 8, 0):
            idx.set_option("zero_copy_batch", zc)
            res[zc] = p50(lambda: idx.search_batch(Q, k) if B > 1 else idx.search(Q[0], k))
        print(f"rows={rows} B={B:3d}: device-resident call + sync {t_dev:.4f} ms; host API: zero-copy up to 256: {res[256]:.4f}, up to 8: {res[8]:.4f}, "
              f"never: {res[0]:.4f} ms", flush=True)
    idx.close()
