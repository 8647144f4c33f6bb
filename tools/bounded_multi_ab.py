"""Bounded exact pass of BATCHES, A/B of one process-wide knob at a time (dev tool; edit the loop for another knob): here option
"bounded_multi_packed" 0 / 1 — the int8 or the packed 5-bit shadow (earlier runs of the same loop: fragment ring 6 / 12 with
"bounded_ring", 4 / 8 waves per workgroup with "bounded_multi_waves": profiles/r04/bounded_ring_ab_100M.log,
bounded_multi_waves_100M.log).  Topical rows (synth_dist 4), queries = further rows of the same stream; a batch of 256 without the
batch feedback (equal ladder share in both arms) and single queries sent to the bounded pass directly ("ladder_feedback" = 2).
python tools/bounded_multi_ab.py [rows=100000000] [rounds=2]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
QROW0 = 1 << 40
qi = dawn.VectorIndex(0)
qi.set_option("synth_dist", 4)
qi.fill_synthetic(1, QROW0, 256 * 256, 1)
Q = qi.get_rows(0, 256 * 256)[0][::256].copy()
qi.close()
idx = dawn.VectorIndex(0)
idx.set_option("synth_dist", 4)
idx.fill_synthetic(1, 0, rows, 1)
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
k = 10
blob = torch.zeros((dawn.result_blob_bytes(256, k),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()
ref = {}
for r in range(rounds):
    for ring in (0, 1):  # (the batch form on the int8 / the packed 5-bit shadow; earlier logs: fragment ring 6 vs 12, 4 vs 8 waves)
        idx.set_option("bounded_multi_packed", ring)
        for name, B, fb, iters in (("batch 256", 256, 0, 6), ("single queries, bounded pass directly", 1, 2, 32)):
            idx.set_option("ladder_feedback", fb)
            for _ in range(2):
                idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()
            s0 = idx.stats()
            t0 = time.perf_counter()
            for i in range(iters):
                qp = d_q.data_ptr() + (0 if B > 1 else (i % 256) * 384 * 4)
                idx.search_device(qp, B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / iters * 1e3
            s1 = idx.stats()
            raw = blob.cpu().numpy()[:B * k * 12].copy()
            same = np.array_equal(ref.setdefault(name, raw), raw)
            print(f"rows={rows} multi_packed={ring:2d} {name:40s}: {el:8.3f} ms per search; bounded per query "
                  f"{(s1['bounded'] - s0['bounded']) / (iters * B):.3f}; identical: {same}", flush=True)
