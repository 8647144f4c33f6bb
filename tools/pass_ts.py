"""When does every wave of the batched int8 pass (scan_i8_pipe16_kernel) leave the kernel (experiments library:
DAWN_LIB=.../libdawn_hip_exp.so; dev tool) — is there a tail of slow workgroups under the static tile assignment?
python tools/pass_ts.py [rows=100000000] [B=256]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import _lib, synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(3, 0, B)
for r in range(5):
    idx.search_batch(Q, 10)
    ts = (C.c_ulonglong * 1024)()
    assert _lib.lib.dawn_debug_read_ts_pass(ts, 1024) == 0
    t = np.array(list(ts), dtype=np.float64).reshape(256, 4) / 100.0  # us
    t -= t.min()
    wg = t.max(axis=1)
    byx = [wg[np.arange(256) % 8 == x].mean() for x in range(8)]
    print(f"batch {r}: last wave leaves {t.max():7.1f} us after the first; workgroups: p10 {np.percentile(wg, 10):7.1f} median {np.median(wg):7.1f} "
          f"p90 {np.percentile(wg, 90):7.1f}; mean per XCD: " + " ".join(f"{v:7.1f}" for v in byx), flush=True)
