#!/usr/bin/env python3
"""ONE process driving N GPUs through the C ABI: dawn_index_create_sharded + the unchanged dawn_index_* calls (what the
Rust drop-in of INTEGRATION.md §2b binds).  Prints one JSON line per gather mode as soon as it is measured:
  {"mode": "peer_copies"|"rccl_all_gather", "n_gpus": N, "rows": R, "batch1": {...}, "batch256": {...}}
bench.py runs this as a child of rank 0 (after the ranks have released their indexes) when N > 1 and folds the lines into
`extra.single_process_sharded`; on a 1-GPU box `--logical G` deals the rows over G shards of device 0 instead."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--logical", type=int, default=0, help="G shards on device 0 (1-GPU boxes)")
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import dawnsearch_amd as dawn
    from dawnsearch_amd import synth

    devices = [0] * args.logical if args.logical else list(range(args.gpus))
    # (gather mode, issuing threads, name)
    modes = [(2, 1, "peer_copies"), (2, 0, "peer_copies_single_issuing_thread")]
    if not args.logical:
        modes.append((1, 1, "rccl_all_gather"))
    idx = dawn.VectorIndex(devices=devices)
    t0 = time.time()
    idx.fill_synthetic(1, 0, args.rows, 1)
    fill_s = time.time() - t0
    q1 = synth.planted_queries(1, [4242 % args.rows], 5)
    Q = synth.unit_rows(3, 0, 256)
    Q[0] = q1[0]
    for mode, threads, name in modes:
        out = {"mode": name, "issuing_threads_per_shard": bool(threads), "n_gpus": len(devices), "logical_shards_on_one_device": bool(args.logical),
               "rows": args.rows, "k": args.k, "fill_seconds": fill_s}
        try:
            idx.set_option("shard_gather", mode)
            idx.set_option("shard_threads", threads)
            for B, qs in ((1, q1), (256, Q)):
                steps = args.steps if B == 1 else max(5, args.steps // 3)
                for _ in range(3):
                    lab, dist, found = idx.search_batch(qs, args.k)
                lat = []
                t0 = time.perf_counter()
                for _ in range(steps):
                    t1 = time.perf_counter()
                    lab, dist, found = idx.search_batch(qs, args.k)
                    lat.append(time.perf_counter() - t1)
                el = time.perf_counter() - t0
                lat = np.array(lat) * 1e3
                out[f"batch{B}"] = {"queries_per_s": steps * B / el, "ms_per_call_mean": el / steps * 1e3,
                                    "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)),
                                    "planted_top1_ok": bool(lab[0][0] == 1 + (4242 % args.rows)), "steps": steps,
                                    "planted_labels": [int(v) for v in lab[0]],
                                    "timing": "host API (H2D queries, N shard searches, gather, merge, D2H results, sync)"}
            out["shard_info"] = idx.shard_info()
            out["stats"] = idx.stats()
        except Exception as e:  # keep what was measured; the parent reports the failure
            out["error"] = repr(e)
        print(json.dumps(out), flush=True)
    idx.close()


if __name__ == "__main__":
    main()
