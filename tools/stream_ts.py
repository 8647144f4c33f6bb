"""Phase timestamps of scan_filter_i8s_kernel (experiments library: DAWN_LIB=.../libdawn_hip_exp.so; dev tool): where the
~20 us of fixed cost of the batch-1 int8 stream go.  Workgroups 0 / 85 / 170 / 255, first and last wave."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth, _lib

names = ["entry", "query images made (barrier)", "B fragments in registers", "1st sub-tile done", "2nd sub-tile done", "stream done",
         "block merge done", "list written"]
for n in (4096, 262_144, 1_000_000, 4_000_000):
    idx = dawn.VectorIndex(0); idx.fill_synthetic(1, 0, n, 1)
    idx.set_option("shadow_scan_unroll", 3); idx.set_option("shadow_scan_threads", 512); idx.set_option("shadow_scan_blocks", 256)
    Q = synth.unit_rows(2, 0, 4)
    reps = 30
    acc = np.zeros((8, 8))
    for r in range(reps):
        idx.search(Q[r % 4], 10)
        ts = (C.c_ulonglong * 64)()
        assert _lib.lib.dawn_debug_read_ts_i8(ts, 64) == 0
        t = np.array(list(ts), dtype=np.float64).reshape(8, 8)
        t0 = t[:, 0].min()
        acc += (t - t0) * 10.0 / 1000.0  # 100-MHz ticks -> us, relative to the earliest entry of the sampled waves
    acc /= reps
    print("rows", n, "(us since the first sampled wave entered the kernel; columns: workgroup 0 / 85 / 170 / 255 x first / last wave)")
    for i, name in enumerate(names):
        print(f"   {name:32s}", "  ".join(f"{acc[w, i]:7.2f}" for w in range(8)))
