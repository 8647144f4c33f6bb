"""Per-leg kernel trace (dev tool).  Two modes:
  run:      rocprofv3 --kernel-trace -d DIR -o legs -- python3 tools/leg_trace.py run [rows,rows,...]
            device-resident searches (batch 1, then batch 256) on an index of each size, a marker kernel (iota of a distinct length)
            between the legs
  summary:  python3 tools/leg_trace.py summary <results.db>
            the kernel sequence of ONE steady-state search of every leg: duration of each kernel, gap in front of it, and the sum
            against the leg's wall time per search
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(sizes):
    import numpy as np
    import torch

    import dawnsearch_amd as dawn
    from dawnsearch_amd import synth
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    Q = synth.unit_rows(2, 0, 256)
    d_q = torch.from_numpy(Q).to(dev)
    for rows in sizes:
        idx = dawn.VectorIndex(0)
        idx.fill_synthetic(1, 0, rows, 1)
        for B, iters in ((1, 60), (256, 20)):
            blob = torch.zeros((dawn.result_blob_bytes(B, 10),), dtype=torch.uint8, device=dev)
            p = blob.data_ptr()
            for _ in range(5):
                idx.search_device(d_q.data_ptr(), B, 10, p, p + B * 80, p + B * 120, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                idx.search_device(d_q.data_ptr(), B, 10, p, p + B * 80, p + B * 120, stream)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / iters * 1e3
            print(f"LEG rows={rows} batch={B} ms_per_search={el:.4f} iters={iters}", flush=True)
        idx.close()


def summary(db):
    import re
    import sqlite3
    c = sqlite3.connect(db)
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    rows = [dict(zip(cols, r)) for r in c.execute("select * from kernels order by start")]

    def short(n):
        m = re.search(r"dawn::(\w+(?:<[^>]*>)?)", n) or re.search(r"_ZN4dawn\d+(\w+?)I", n)
        return m.group(1) if m else n[:40]
    seq = [(short(r["name"]), r["start"], r["end"], r.get("grid_x")) for r in rows]
    # legs: a steady-state search = the kernels between two consecutive launches of the same "first kernel"; legs are cut at the
    # fill kernels (synth_write_kernel)
    cuts = [i for i, s in enumerate(seq) if s[0].startswith("synth_write")] + [len(seq)]
    starts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        seg = seq[a + 1:b]
        # (torch's own kernels — the fills behind torch.zeros, copies — are not part of a search)
        seg = [s for s in seg if not s[0].startswith(("rows_to_", "synth_", "iota", "validate", "__amd_rocclr", "void at::", "at::"))]
        if not seg:
            continue
        # split into the batch-1 part and the batch-256 part: the latter starts at the first prep_queries_i8 launch
        i256 = next((i for i, s in enumerate(seg) if s[0].startswith("prep_queries_i8")), len(seg))
        for name, part in (("batch 1", seg[:i256]), ("batch 256", seg[i256:])):
            if len(part) < 8:
                continue
            first = part[0][0]
            idxs = [i for i, s in enumerate(part) if s[0] == first]
            if len(idxs) < 4:
                continue
            i0, i1 = idxs[-3], idxs[-2]
            one = part[i0:i1]
            span = (part[i1][1] - part[i0][1]) / 1e3
            print(f"--- leg after fill #{cuts.index(a) + 1}, {name}: one search = {len(one)} kernels, start to next start {span:.1f} us")
            for j, (n, s, e, g) in enumerate(one):
                gap = (s - part[i0 + j - 1][2]) / 1e3 if (i0 + j) > 0 else 0.0
                print(f"    {n:44s} grid {str(g):>8s}  dur {(e - s) / 1e3:9.2f} us   gap before {gap:6.2f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1_000_000, 12_500_000]
        run(sizes)
    else:
        summary(sys.argv[2])
