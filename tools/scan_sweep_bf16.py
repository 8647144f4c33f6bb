"""Sweep batch-1 scan geometry for the bf16 index (dev tool): python tools/scan_sweep_bf16.py [rows] [iters]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
idx = dawn.VectorIndex(0, dtype="bf16")
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 8)
bytes_per = rows * 768
res = []
for unroll in (1, 2):
    for threads in (128, 256, 512, 1024):
        for blocks in (256, 512, 1024):
            if blocks * threads > 256 * 2048 or blocks * threads < 256 * 256:
                continue
            idx.set_option("scan_unroll", unroll); idx.set_option("scan_threads", threads); idx.set_option("scan_blocks", blocks)
            idx.search_batch(Q[:1], 10)
            idx.profile_enable(True)
            for _ in range(iters):
                idx.search_batch(Q[:1], 10)
            n, ms = idx.profile_read()
            idx.profile_enable(False)
            k_ms = ms / max(n, 1)
            res.append((bytes_per / k_ms / 1e6, unroll, threads, blocks))
            print(f"U={unroll} threads={threads:4d} blocks={blocks:4d}  scan {k_ms*1e3:9.1f} us  {bytes_per/k_ms/1e6:8.1f} GB/s", flush=True)
res.sort(reverse=True)
print("best:", res[:6])
