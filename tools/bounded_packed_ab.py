"""Bounded exact pass of single queries: the packed 5-bit shadow (240 B/row, option "bounded_packed" 1) against the int8 shadow
(384 B/row, 0) — dev tool.  Topical rows (synth_dist 4 / 5), queries = further rows of the same stream: every query sent to the
bounded pass directly ("ladder_feedback" = 2), and the default ladder (packed stream first, feedback on).  Answers compared bit for bit.
python tools/bounded_packed_ab.py [rows=100000000] [dist=4] [queries=96]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dist = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dev = torch.device("cuda", 0)
QROW0 = 1 << 40
qi = dawn.VectorIndex(0)
qi.set_option("synth_dist", dist)
qi.fill_synthetic(1, QROW0, nq * 256, 1)
Q = qi.get_rows(0, nq * 256)[0][::256].copy()
qi.close()
idx = dawn.VectorIndex(0)
idx.set_option("synth_dist", dist)
idx.fill_synthetic(1, 0, rows, 1)
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
ref = {}
for k in (10, 20):
    blob = torch.zeros((nq, dawn.result_blob_bytes(1, k)), dtype=torch.uint8, device=dev)
    shifts = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else []  # (seed fraction sweep: bounded_seed_shift)
    forced = len(sys.argv) > 4 and sys.argv[4] == "forced"  # (below 40 Mi rows the defaults keep the int8 shadow: force the other forms)
    for packed, seed in (((0, 0), (2, 0), (2, 2), (0, 0), (2, 0), (2, 2)) if forced else ((0, 0), (1, 0), (1, 1), (0, 0), (1, 0), (1, 1))):
        idx.set_option("bounded_packed", packed)
        idx.set_option("bounded_seed", seed)
        for name, fb in (("bounded pass directly", 2), ("default ladder (packed stream first, feedback)", 1),
                         ("packed stream first, no feedback", 0)):
            idx.set_option("ladder_feedback", fb)
            for sh in (shifts if (seed and fb == 2 and shifts) else [5]):
              idx.set_option("bounded_seed_shift", sh)
              name_sh = name if sh == 5 and not shifts else f"{name}, seed over n >> {sh}"
              lat = []
              s0 = idx.stats()
              for rep in range(2):
                  for i in range(nq):
                      p = blob[i].data_ptr()
                      torch.cuda.synchronize()
                      t0 = time.perf_counter()
                      idx.search_device(d_q.data_ptr() + i * 384 * 4, 1, k, p, p + k * 8, p + k * 12, stream)
                      torch.cuda.synchronize()
                      if rep:
                          lat.append((time.perf_counter() - t0) * 1e3)
              s1 = idx.stats()
              raw = blob.cpu().numpy()[:, :k * 12].copy()
              same = np.array_equal(ref.setdefault(k, raw), raw)
              lat = np.array(lat)
              print(f"rows={rows} dist={dist} k={k} bounded_packed={packed} seed={seed} {name_sh:48s}: mean {lat.mean():6.3f} p50 {np.percentile(lat, 50):6.3f} "
                    f"p95 {np.percentile(lat, 95):6.3f} max {lat.max():6.3f} ms; bounded {(s1['bounded'] - s0['bounded']) / (2 * nq):.2f} "
                    f"demoted {(s1['demoted'] - s0['demoted']) / (2 * nq):.2f} packed failures {(s1['packed_failures'] - s0['packed_failures']) / (2 * nq):.2f} fallbacks {s1['fallbacks'] - s0['fallbacks']}; identical: {same}",
                    flush=True)
