"""Run a few batched searches (for rocprofv3): python tools/prof_batched.py [rows] [B] [waves] [iters] [dtype] [mfma_sched]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 8
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
idx = dawn.VectorIndex(0, dtype=dtype)
idx.fill_synthetic(1, 0, rows, 1)
if len(sys.argv) > 6:
    idx.set_option("mfma_sched", int(sys.argv[6]))
Q = synth.unit_rows(2, 0, B)
for _ in range(iters):
    idx.search_batch(Q, 10)
print(idx.stats())
