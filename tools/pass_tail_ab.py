"""A/B of the int8 matrix-core pass's tile assignment (dev tool): static for every tile (option mfma_dynamic_tail = 0) against the
last eighth handed out on demand (default), interleaved rounds on one index, results compared bit for bit.
python tools/pass_tail_ab.py [rows=100000000] [B=256] [rounds=4]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(3, 0, B)
ref = None
res = {0: [], 1: []}
for r in range(rounds):
    for mode in (1, 0):
        idx.set_option("mfma_dynamic_tail", mode)
        out = idx.search_batch(Q, 20)
        ref = ref or out
        assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1].view(np.uint32), ref[1].view(np.uint32)), mode
        ms = idx.debug_time_full_pass(B, 8) * 1e3
        t0 = time.perf_counter()
        for _ in range(5):
            idx.search_batch(Q, 20)
        res[mode].append((ms, (time.perf_counter() - t0) / 5 * 1e3))
for mode, name in ((0, "static"), (1, "dynamic tail")):
    print(f"rows={rows} B={B} {name:13s}: full pass us {[round(a, 1) for a, _ in res[mode]]}  search ms {[round(b, 3) for _, b in res[mode]]}")
print(idx.stats())
