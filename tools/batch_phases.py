"""Where a wave of the batched scan kernel spends its cycles (diagnostic build, mfma_sched = 2):
python tools/batch_phases.py [rows] [B] [dtype]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
from dawnsearch_amd._lib import lib, check
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
idx = dawn.VectorIndex(0, dtype=dtype)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(2, 0, B)
idx.search_batch(Q, 10)
mode = 2
idx.set_option("mfma_sched", mode)
idx.search_batch(Q, 10)
out = np.zeros((256, 8, 8), dtype=np.uint64)
check(lib.dawn_index_debug_read_diag(idx._h, C.c_void_p(out.ctypes.data), 256))
idx.set_option("mfma_sched", 0)
names = ["wait loads + convert + issue", "barrier", "contract sub0", "epilogue sub0", "contract sub1", "epilogue sub1"]
tiles = -(-rows // 64) / 256.0
tot = out[:, :, :6].sum(axis=2).mean()
print(f"{dtype} rows={rows} B={B}: {tiles:.0f} tiles per workgroup, {tot / tiles:.0f} stamped cycles per tile per wave")
for k, n in enumerate(names):
    v = out[:, :, k].astype(np.float64)
    print(f"  {n:30s} {v.mean() / tiles:8.0f} cycles/tile  {100 * v.mean() / tot:5.1f} %   (waves 0-3: {v[:, :4].mean() / tiles:6.0f}, waves 4-7: {v[:, 4:].mean() / tiles:6.0f})")
