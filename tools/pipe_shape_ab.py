"""A/B of the int8 batched filter's MFMA shape (dev tool): option mfma_sched 4 = v_mfma_i32_16x16x64_i8 (default), 32 =
v_mfma_i32_32x32x32_i8.  usage: pipe_shape_ab.py [rows ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dawnsearch_amd as dawn
from dawnsearch_amd import synth

rows_list = [int(a) for a in sys.argv[1:]] or [1_000_000, 100_000_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for rows in rows_list:
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, rows, 1)
    for B in (256, 64, 16):
        Q = synth.unit_rows(2, 0, B)
        dq = torch.from_numpy(Q).to(dev)
        lab = torch.zeros((B, 10), dtype=torch.int64, device=dev)
        dist = torch.zeros((B, 10), dtype=torch.float32, device=dev)
        found = torch.zeros((B,), dtype=torch.int32, device=dev)
        res = {}
        for sched in (32, 4, 32, 4):
            idx.set_option("mfma_sched", sched)
            n = 50 if rows <= 10_000_000 else 5
            for _ in range(2):
                idx.search_device(dq.data_ptr(), B, 10, lab.data_ptr(), dist.data_ptr(), found.data_ptr(), stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                idx.search_device(dq.data_ptr(), B, 10, lab.data_ptr(), dist.data_ptr(), found.data_ptr(), stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            res.setdefault(sched, []).append(ms)
            out = lab.cpu().numpy().copy()
            if sched == 32 and "ref" not in res:
                res["ref"] = out
            else:
                assert np.array_equal(out, res["ref"]), "results differ between the two MFMA shapes"
        print(f"rows {rows:>11,d} B={B:3d}: 32x32x32 {res[32][0]:.3f} / {res[32][1]:.3f} ms   16x16x64 {res[4][0]:.3f} / {res[4][1]:.3f} ms   st {idx.stats()}", flush=True)
    idx.close()
