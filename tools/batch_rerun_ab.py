"""Second pass of a batch's flagged queries with exact-derived thresholds (option "batch_rerun") against the bounded pass alone —
dev tool.  Topical rows, a batch of 256 further rows of the same stream, the sampled thresholds at the default depth (1024) and at
the depth the batch feedback chooses for such an index (4096).  python tools/batch_rerun_ab.py [rows=100000000] [dists=4,5]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dists = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4, 5]
dev = torch.device("cuda", 0)
QROW0 = 1 << 40
for dist in dists:
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    qi.fill_synthetic(1, QROW0, 256 * 256, 1)
    Q = qi.get_rows(0, 256 * 256)[0][::256].copy()
    qi.close()
    idx = dawn.VectorIndex(0)
    idx.set_option("synth_dist", dist)
    idx.fill_synthetic(1, 0, rows, 1)
    idx.set_option("ladder_feedback", 0)  # (the depth is set by hand below)
    d_q = torch.from_numpy(Q).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    for k in (10, 20):
        blob = torch.zeros((dawn.result_blob_bytes(256, k),), dtype=torch.uint8, device=dev)
        p = blob.data_ptr()
        ref = None
        for target in (1024, 4096):
            idx.set_option("mfma_target", target)
            for rerun in (0, 2):
                idx.set_option("batch_rerun", rerun)
                for _ in range(2):
                    idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
                torch.cuda.synchronize()
                s0, f0 = idx.stats(), idx.stats_batch_feedback()
                t0 = time.perf_counter()
                for _ in range(5):
                    idx.search_device(d_q.data_ptr(), 256, k, p, p + 256 * k * 8, p + 256 * k * 12, stream)
                torch.cuda.synchronize()
                el = (time.perf_counter() - t0) / 5 * 1e3
                s1, f1 = idx.stats(), idx.stats_batch_feedback()
                raw = blob.cpu().numpy()[:256 * k * 12].copy()
                if ref is None:
                    ref = raw
                print(f"rows={rows} dist={dist} k={k} target={target} batch_rerun={rerun}: {el:7.2f} ms per batch of 256; per query: second pass "
                      f"answered {(f1['rerun_answers'] - f0['rerun_answers']) / 1280:.3f}, bounded {(s1['bounded'] - s0['bounded']) / 1280:.3f}, "
                      f"fallbacks {s1['fallbacks'] - s0['fallbacks']}; identical: {np.array_equal(ref, raw)}", flush=True)
