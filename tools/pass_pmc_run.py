"""Batched passes for a PMC run (dev tool): batches of 256 queries on an index of `rows` rows through the int8 pass, then through
the FP6 first filter.  rocprofv3 --kernel-trace --pmc <counters> -d DIR -o pass -- python3 tools/pass_pmc_run.py [rows=12500000];
summary: python3 tools/pmc_summary.py <results.db> 300"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dev = torch.device("cuda", 0)
idx = dawn.VectorIndex(0)
idx.set_option("f6_min_rows", 0)
idx.fill_synthetic(1, 0, rows, 1)
Q = synth.unit_rows(3, 0, 256)
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
blob = torch.zeros((dawn.result_blob_bytes(256, 10),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
for f6 in (0, 1):
    idx.set_option("f6_shadow", f6)
    for _ in range(12):
        idx.search_device(d_q.data_ptr(), 256, 10, p, p + 256 * 80, p + 256 * 120, stream)
    torch.cuda.synchronize()
print(idx.stats())
