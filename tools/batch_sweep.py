"""Batched-scan timing on the GPU box (dev tool): python tools/batch_sweep.py [rows] [B,B,...] [dtype]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Bs = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [9, 64, 256]
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
idx = dawn.VectorIndex(0, dtype=dtype)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 256)
rb = 768 if dtype == "bf16" else 1536
for B in Bs:
    for waves, sched in ([(4, int(x)) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else ((8, 0), (8, 1), (4, 4))):
        idx.set_option("mfma_sched", sched)
        idx.search_batch(Q[:B], 10)
        idx.profile_enable(True)
        t0 = time.time(); it = 10 if sched < 40 else 2
        for _ in range(it): idx.search_batch(Q[:B], 10)
        wall = (time.time() - t0) / it
        n, ms = idx.profile_read(); idx.profile_enable(False)
        k_ms = ms / max(n, 1)
        flops = 2.0 * 256 * rows * 384
        print(f"{dtype} rows={rows} B={B:3d} waves={waves} sched={sched} scan {k_ms*1e3:9.1f} us  {flops/k_ms/1e9:7.1f} TFLOP/s(256q)  {rows*rb/k_ms/1e6:7.1f} GB/s  wall {wall*1e3:8.3f} ms  qps {B/wall:9.0f}", flush=True)
print(idx.stats())
