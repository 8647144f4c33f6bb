"""The wide batch form of the bounded pass for a PMC run (dev tool): batches of 64 queries forced through the ladder on uniform rows.
rocprofv3 --kernel-trace --pmc <counters> -d DIR -o pass -- python3 tools/wide_pmc_run.py [rows=12500000] [dist=0]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
idx = dawn.VectorIndex(0)
idx.set_option("synth_dist", dist)
idx.fill_synthetic(1, 0, rows, 1)
idx.set_option("force_fallback", 2)
idx.set_option("ladder_feedback", 0)
Q = synth.unit_rows(3, 0, 64)
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
blob = torch.zeros((dawn.result_blob_bytes(64, 10),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
for _ in range(12):
    idx.search_device(d_q.data_ptr(), 64, 10, p, p + 64 * 80, p + 64 * 120, stream)
torch.cuda.synchronize()
print(idx.stats_raw())
