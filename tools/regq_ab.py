"""A/B of the batched append pass (dev tool): scan_i8_pipe16_kernel (mfma_sched 4: LDS-DMA tiles, queries of a wave in registers)
against scan_i8_regq_kernel (mfma_sched 6: 128 queries per wave in AGPRs, rows straight from global memory).  Results are
compared bit for bit; the pass kernel is timed with the library's HIP events.  python tools/regq_ab.py [rows] [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
iters = 4 if rows > 20_000_000 else 20 if rows > 2_000_000 else 60
for B in (256, 128, 64, 17):
    Q = synth.unit_rows(3, 0, B)
    Q[0] = synth.planted_queries(1, [rows // 2], 4)[0]
    idx.set_option("mfma_sched", 4)
    want = idx.search_batch(Q, 20)
    acc = {4: [], 6: []}
    for r in range(rounds):
        for sched in (4, 6):
            idx.set_option("mfma_sched", sched)
            got = idx.search_batch(Q, 20)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), (B, sched)
            idx.profile_enable(True)
            for _ in range(iters):
                idx.search_batch(Q, 20)
            n, ms = idx.profile_read()
            idx.profile_enable(False)
            acc[sched].append(ms / n)
    for sched, v in acc.items():
        k = min(v)
        print(f"B={B:3d} mfma_sched={sched}: pass best {k * 1e3:9.1f} us  all {[round(x * 1e3, 1) for x in v]}  "
              f"{2.0 * 256 * rows * 384 / k / 1e9 / 1e3:7.1f} Top/s(256 cols)  {rows * 384.25 / k / 1e6:7.1f} GB/s", flush=True)
idx.set_option("mfma_sched", 4)
print(idx.stats())
