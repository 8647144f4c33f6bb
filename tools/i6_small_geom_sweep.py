"""Geometry of the packed stream on small indexes (dev tool): ms per single-query search vs waves per workgroup and loads in flight —
fewer waves refine fewer rows (every wave re-scores its own list: 2048 waves x 24 rows x 1.5 KB = 75 MB next to a 240-MB stream of
1 M rows).   python tools/i6_small_geom_sweep.py [rows ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 2_000_000, 4_000_000, 12_500_000]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
NQ = 256
Q = synth.unit_rows(11, 0, NQ)
d_q = torch.from_numpy(Q).to(dev)
for rows in sizes:
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, rows, 1)
    idx.set_option("ladder_feedback", 0)
    k = 10
    blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
    p = blob.data_ptr()

    def timed(label):
        for i in range(8):
            idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
        torch.cuda.synchronize()
        r0 = idx.stats_raw()
        t0 = time.perf_counter()
        for rep in range(2):
            for i in range(NQ):
                idx.search_device(d_q.data_ptr() + i * 1536, 1, k, p, p + k * 8, p + k * 12, stream)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / (2 * NQ) * 1e3
        r1 = idx.stats_raw()
        print(f"rows={rows} {label}: {el:.4f} ms per search, lists of {idx.i6_refine(k)[0]}, ladder {[b - a for a, b in zip(r0, r1)][1:6]}", flush=True)

    timed("default geometry          ")
    for threads in (128, 256, 384, 512):
        for ring in (4, 8):
            for chunk, share in ((8, 3), (16, 2)):
                idx.set_option("i6_scan_threads", threads)
                idx.set_option("i6_scan_ring", ring)
                idx.set_option("i6_dyn_chunk", chunk)
                idx.set_option("i6_dyn_share", share)
                timed(f"threads={threads} ring={ring} chunk={chunk:2d}/{share}")
    idx.close()
