"""Dev tool: time the matrix-core full pass alone for several mfma_sched variants (41..55 = parts switched off, wrong
results but discarded): python tools/pass_ablation.py [rows] [B] [variants]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 4, 41, 42, 43, 44, 47, 48, 55]
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, B)
idx.search_batch(Q, 10)  # thresholds
for rep in range(2):
    for v in variants:
        idx.set_option("mfma_sched", v)
        idx.debug_time_full_pass(B, 2)
        ms = idx.debug_time_full_pass(B, 8)
        print(f"rep {rep} sched={v:3d}  full pass {ms*1e3:9.1f} us  {2.0*256*rows*384/ms/1e9:7.1f} TFLOP/s(256q)", flush=True)
