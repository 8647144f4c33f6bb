"""The packed stream's list sizing against many queries (dev tool): for each size and k — the depth i6_refine_count chooses from the
shadow's measured error bounds (and from the constants: "i6_slack_model" 0), ms per search with either, and how many of NQ distinct
uniform queries lose their certificate.   python tools/refine_check.py [NQ=1024] [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sizes = [int(a) for a in sys.argv[2:]] or [2_200_000, 12_500_000, 50_000_000, 100_000_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
Q = synth.unit_rows(11, 0, NQ)
d_q = torch.from_numpy(Q).to(dev)
for rows in sizes:
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, rows, 1)
    idx.set_option("ladder_feedback", 0)
    n10, frac = idx.i6_refine(10)
    nz = [(b, f) for b, f in enumerate(frac) if f > 0]
    mean_e = sum((b + 0.5) * 0.004 * f for b, f in nz)
    print(f"rows={rows}: E histogram bins {nz[0][0] * 0.004:.3f}..{(nz[-1][0] + 1) * 0.004:.3f}, mean ~{mean_e:.4f}", flush=True)
    for k in (1, 10, 20, 64):
        blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
        p = blob.data_ptr()
        for model in (1, 0):
            idx.set_option("i6_slack_model", model)
            n = idx.i6_refine(k)[0]
            nq = NQ if rows <= 50_000_000 else NQ // 2
            for i in range(4):
                idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
            torch.cuda.synchronize()
            r0 = idx.stats_raw()
            t0 = time.perf_counter()
            for i in range(nq):
                idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / nq * 1e3
            r1 = idx.stats_raw()
            d = [b - a for a, b in zip(r0, r1)]
            print(f"rows={rows} k={k:2d} {'measured' if model else 'constants'}: lists of {n:2d}: {el:7.4f} ms per search; of {nq}: "
                  f"certificates lost {d[5]}, second chance {d[2]}, bounded {d[4]}, exact {d[1]}", flush=True)
    idx.close()
    torch.cuda.synchronize()
    time.sleep(1.5)
