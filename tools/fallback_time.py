"""Dev tool: cost of the exact fallback pass (forced for every query): python tools/fallback_time.py [rows] [dtype]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
idx = dawn.VectorIndex(0, dtype=dtype)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, 16)
for B in (1, 4, 16):
    idx.set_option("force_fallback", 0)
    idx.search_batch(Q[:B], 10)
    t0 = time.time(); idx.search_batch(Q[:B], 10); t_ok = time.time() - t0
    idx.set_option("force_fallback", 1)
    idx.search_batch(Q[:B], 10)
    t0 = time.time(); idx.search_batch(Q[:B], 10); t_fb = time.time() - t0
    print(f"{dtype} rows={rows} B={B:2d}: normal {t_ok*1e3:9.2f} ms   with the exact pass {t_fb*1e3:9.2f} ms  (+{(t_fb-t_ok)/B*1e3:8.2f} ms per query)", flush=True)
