"""A/B of the batched search's first filter (dev tool): the int8 pass (default) against the FP6 (e2m3) shadow + int8 refinement
(option f6_shadow = 1), interleaved rounds, device-resident batches between two synchronisations, the dominant kernel's own
duration from its HIP events, answers compared bit for bit, certificate counters.
python tools/f6_batch_ab.py [rows=100000000] [rounds=3] [batch=256] [targets=12288]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
targets = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [12288]
dist = int(sys.argv[5]) if len(sys.argv) > 5 else 0
staggers = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else [8]
dev = torch.device("cuda", 0)
idx = dawn.VectorIndex(0)
if dist:
    idx.set_option("synth_dist", dist)
idx.set_option("f6_min_rows", 0)
t0 = time.perf_counter()
idx.fill_synthetic(1, 0, rows, 1)
print(f"rows={rows} fill {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
idx.set_option("f6_shadow", 1)
print(f"FP6 shadow built in {time.perf_counter() - t0:.2f} s; memory {idx.memory()}", flush=True)
if dist >= 4:
    qi = dawn.VectorIndex(0)
    qi.set_option("synth_dist", dist)
    qi.fill_synthetic(1, 1 << 40, 256 * 256, 1)
    Q = qi.get_rows(0, 256 * 256)[0][::256].copy()[:B]
    qi.close()
else:
    Q = synth.unit_rows(3, 0, B)
    Q[0] = synth.planted_queries(1, [4242 % rows], 5)[0]
d_q = torch.from_numpy(Q).to(dev)
stream = torch.cuda.current_stream().cuda_stream
iters = 8 if rows > 30_000_000 else 30
for k in (10, 20):
    nb = dawn.result_blob_bytes(B, k)
    blob = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    p = blob.data_ptr()
    cfgs = [("int8 pass", 0, 0, 0)] + [(f"FP6 first filter, target {t}, stagger {sg}", 1, t, sg) for t in targets for sg in staggers]
    res = {c[0]: [] for c in cfgs}
    ker = {c[0]: [] for c in cfgs}
    st = {}
    ans = {}
    for r in range(rounds):
        for name, f6, tgt, sg in cfgs:
            idx.set_option("f6_shadow", f6)
            if f6:
                idx.set_option("f6_target", tgt)
                idx.set_option("f6_stagger", sg)
            for _ in range(2):
                idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()
            s0 = idx.stats()
            idx.profile_enable(True)
            t0 = time.perf_counter()
            for _ in range(iters):
                idx.search_device(d_q.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, stream)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            n, ms = idx.profile_read()
            idx.profile_enable(False)
            s1 = idx.stats()
            res[name].append(el / iters * 1e3)
            ker[name].append(ms / max(n, 1))
            st[name] = {kk: (s1[kk] - s0[kk]) / (iters * B) for kk in ("second_chances", "deepened", "bounded", "fallbacks")}
            raw = blob.cpu().numpy()
            ans[name] = (raw[:B * k * 8].copy(), raw[B * k * 8:B * k * 12].copy())
    base = ans["int8 pass"]
    for name, _, _, _ in cfgs:
        same = np.array_equal(ans[name][0], base[0]) and np.array_equal(ans[name][1], base[1])
        print(f"rows={rows} B={B} k={k} {name:46s}: ms per batch {[round(v, 3) for v in res[name]]} best {min(res[name]):.3f}; pass kernel ms "
              f"{[round(v, 3) for v in ker[name]]}; rates {st[name]}; identical to the int8 pass: {same}", flush=True)
