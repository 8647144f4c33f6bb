"""Stream regression A/B (round-4 verdict item 5): the SAME measurement against the libraries of the end of round 2 (9b8dba4), the end
of round 3 (2eecaf7) and this tree, each through its own python package (`git archive <commit> dawnsearch_amd include oracle | tar -x -C
ab_trees/<name>` + make there), on one box, in one run: the batch-1 search streaming the f32 rows (scan_filter_kernel<1,3>: the only
kernel whose bytes are SURVEY 8(d)'s literal N x 384 x 4) and the int8 shadow (scan_filter_i8s_pipe_kernel), each with
"stream_dynamic_tail" 0 / 1 where the tree knows the option.  Kernel time = the library's own HIP events around the scan kernel.
python tools/stream_regression_ab.py <tree> [rows=100000000] [rounds=3]    (one tree per process: the trees' packages share a name)"""
import json
import os
import sys
import time

tree = os.path.abspath(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sys.path.insert(0, tree)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

assert os.path.abspath(dawn.__file__).startswith(tree), dawn.__file__
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
q = torch.from_numpy(synth.unit_rows(2, 0, 1)).to(dev)
k = 10
blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()


SETTLE = float(os.environ.get("SETTLE", "2.0"))  # seconds of idle after an option that may release a shadow (0: as rounds 3 / 4 measured)


def opt(name, v):
    """An option that releases tens of GB (the packed shadow goes when it is not wanted) leaves every stream ~3 % slower until the device
    has idled for a second or two (tools/stream_alloc_probe.py: any hipFree of 24 GB does) — settle before timing."""
    try:
        idx.set_option(name, v)
        torch.cuda.synchronize()
        if SETTLE > 0:
            time.sleep(SETTLE)
        return True
    except Exception:
        return False


def leg(steps=12, warm=3):
    idx.profile_enable(True)
    for _ in range(warm):
        idx.search_device(q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)
    torch.cuda.synchronize()
    idx.profile_read()
    t0 = time.perf_counter()
    for _ in range(steps):
        idx.search_device(q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps * 1e3
    n, ms = idx.profile_read()
    idx.profile_enable(False)
    return el, ms / max(n, 1)


name = os.path.basename(tree.rstrip("/")) if "ab_trees" in tree else "HEAD"
for r in range(rounds):
    for dyn in (1, 0):
        has_dyn = opt("stream_dynamic_tail", dyn)
        if not has_dyn and dyn == 0:
            continue
        # f32 rows: both shadows off for single queries
        opt("f16_shadow_b1", 0)
        el, km = leg()
        print(json.dumps({"tree": name, "round": r, "stream_dynamic_tail": dyn if has_dyn else None, "stream": "f32 rows (1536 B/row)",
                          "kernel_ms": round(km, 4), "search_ms": round(el, 4), "hbm_frac": round(rows * 1536 / (km * 1e-3) / 8e12, 4)}),
              flush=True)
        opt("f16_shadow_b1", 1)
        # int8 shadow: the packed shadow off where the tree has one
        opt("i6_shadow", 0)
        el, km = leg()
        print(json.dumps({"tree": name, "round": r, "stream_dynamic_tail": dyn if has_dyn else None, "stream": "int8 shadow (384.25 B/row)",
                          "kernel_ms": round(km, 4), "search_ms": round(el, 4), "hbm_frac": round(rows * 384.25 / (km * 1e-3) / 8e12, 4)}),
              flush=True)
        opt("i6_shadow", 1)
    opt("stream_dynamic_tail", 1)
