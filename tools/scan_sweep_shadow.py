"""Sweep the batch-1 geometry of the shadow streaming filters (dev tool): python tools/scan_sweep_shadow.py [rows] [unrolls] [i8|f16]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000
kind = sys.argv[3] if len(sys.argv) > 3 else "i8"
RB = 384.25 if kind == "i8" else 768
idx = dawn.VectorIndex(0)
idx.set_option("i8_shadow", int(kind == "i8"))
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 8)
res = []
for unroll in ([int(u) for u in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 3, 4)):
    for threads in (128, 256, 512):
        for blocks in (256, 512, 1024):
            if blocks * threads > 256 * 512 or blocks * threads < 256 * 128 or blocks > 512:
                continue
            idx.set_option("shadow_scan_unroll", unroll); idx.set_option("shadow_scan_threads", threads); idx.set_option("shadow_scan_blocks", blocks)
            idx.search_batch(Q[:1], 10)
            idx.profile_enable(True)
            for _ in range(6 if rows > 10_000_000 else 50):
                idx.search_batch(Q[:1], 10)
            n, ms = idx.profile_read()
            idx.profile_enable(False)
            k_ms = ms / max(n, 1)
            res.append((rows * RB / k_ms / 1e6, unroll, threads, blocks))
            print(f"U={unroll} threads={threads:4d} blocks={blocks:4d}  scan {k_ms*1e3:9.1f} us  {rows*RB/k_ms/1e6:8.1f} GB/s ({RB} B/row)", flush=True)
res.sort(reverse=True)
print("best:", res[:5], idx.stats())
