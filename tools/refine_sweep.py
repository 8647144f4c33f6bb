"""Single queries on the packed stream vs "i6_refine" (entries of its coarse list a wave refines; 0 = i6_refine_count's choice): ms per
search and how many certificates failed, uniform and topical rows (dev tool).  python tools/refine_sweep.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [12_500_000, 25_000_000, 100_000_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
NQ = 64
for rows in sizes:
    for dist in (0, 4):
        qi = dawn.VectorIndex(0)
        qi.set_option("synth_dist", dist)
        qi.fill_synthetic(1, 1 << 40, NQ * 256, 1)
        Q = qi.get_rows(0, NQ * 256)[0][::256].copy()
        qi.close()
        idx = dawn.VectorIndex(0)
        idx.set_option("synth_dist", dist)
        idx.fill_synthetic(1, 0, rows, 1)
        idx.set_option("ladder_feedback", 0)
        d_q = torch.from_numpy(Q).to(dev)
        for k in (10, 20):
            blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
            p = blob.data_ptr()
            for refine in (0, 8, 16, 24, 32, 40, 48, 64):
                idx.set_option("i6_refine", refine)
                for i in range(4):
                    idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
                torch.cuda.synchronize()
                r0 = idx.stats_raw()
                t0 = time.perf_counter()
                for rep in range(3):
                    for i in range(NQ):
                        idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
                torch.cuda.synchronize()
                el = (time.perf_counter() - t0) / (3 * NQ) * 1e3
                r1 = idx.stats_raw()
                d = [(b - a) / 3 for a, b in zip(r0, r1)]
                print(f"rows={rows} dist={dist} k={k} i6_refine={refine:2d}: {el:7.4f} ms per search; of {NQ}: second chance {d[2]:4.0f} bounded {d[4]:4.0f} "
                      f"exact {d[1]:3.0f}", flush=True)
        idx.close()
        torch.cuda.synchronize()
        time.sleep(1.5)
