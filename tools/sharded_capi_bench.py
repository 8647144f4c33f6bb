#!/usr/bin/env python3
"""ONE process driving N GPUs through the C ABI: dawn_index_create_sharded + the unchanged dawn_index_* calls (what the
Rust drop-in of INTEGRATION.md §2b binds) — the product's own multi-GPU form: row shards, per-shard top-k, ONE grouped
ncclAllGather of the packed per-shard results over xGMI (the library dlopens RCCL itself) or peer copies, merge by
insertion position.  `bench.py --gpus N` runs this as a child of rank 0 BEFORE the ranks build their own indexes and takes
its first line as the HEADLINE of the N-GPU run.

Prints one JSON line per gather mode as soon as it is measured:
  {"mode": "auto"|"rccl_all_gather"|"peer_copies", "n_gpus": N, "rows": R,
   "batch1": {queries_per_s, ms_per_step, p50_ms, p95_ms, ...}, "batch256": {...}, "shard_info": {...}, "stats": {...}}
  * queries_per_s / ms_per_step: W untimed warm-ups, then K back-to-back dawn_index_search_device calls on device-resident
    queries, bracketed by a device synchronisation on both sides (nothing is pipelined by the caller: the next call is issued
    when the previous one has been enqueued);
  * p50_ms / p95_ms: the same call synchronised every time (unpipelined latency: shard searches + gather + merge);
  * host_api_p50_ms: dawn_index_search_batch on host buffers (H2D of the queries and D2H of the results included).
On a 1-GPU box `--logical G` deals the rows over G shards of device 0 instead (functional check of the same code)."""
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
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="batch size of the first (headline) leg")
    ap.add_argument("--modes", default="auto,peer,rccl")
    args = ap.parse_args()
    import torch  # (device buffers for the device-resident calls: plumbing)

    import dawnsearch_amd as dawn
    from dawnsearch_amd import synth

    devices = [0] * args.logical if args.logical else list(range(args.gpus))
    # (shard_gather option, name): 0 = the library's own choice (RCCL when every shard has its own device and RCCL loads)
    table = {"auto": (0, "auto"), "peer": (2, "peer_copies"), "rccl": (1, "rccl_all_gather")}
    modes = [table[m] for m in args.modes.split(",") if m in table]
    if args.logical:
        modes = [m for m in modes if m[0] != 1]  # (RCCL cannot put two ranks on one device)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    idx = dawn.VectorIndex(devices=devices)
    t0 = time.time()
    idx.fill_synthetic(1, 0, args.rows, 1)
    fill_s = time.time() - t0
    q1 = synth.planted_queries(1, [4242 % args.rows], 5)
    Q = synth.unit_rows(3, 0, 256)
    Q[0] = q1[0]
    stream = torch.cuda.current_stream().cuda_stream
    legs = [(args.batch, q1 if args.batch == 1 else Q[:args.batch])]
    legs += [(B, qs) for B, qs in ((1, q1), (256, Q)) if B != args.batch]
    for mode, name in modes:
        out = {"mode": name, "n_gpus": len(devices), "logical_shards_on_one_device": bool(args.logical), "rows": args.rows,
               "k": args.k, "fill_seconds": fill_s}
        try:
            idx.set_option("shard_gather", mode)
            for B, qs in legs:
                steps = args.steps if B == legs[0][0] else max(5, args.steps // 3)
                d_q = torch.from_numpy(np.ascontiguousarray(qs)).to(dev)
                nb = dawn.result_blob_bytes(B, args.k)
                blob = torch.zeros((nb,), dtype=torch.uint8, device=dev)
                p = blob.data_ptr()

                def call():
                    idx.search_device(d_q.data_ptr(), B, args.k, p, p + B * args.k * 8, p + B * args.k * 12, stream)

                for _ in range(args.warmup):
                    call()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    call()
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                lat = []
                for _ in range(min(steps, 100)):
                    t1 = time.perf_counter()
                    call()
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t1)
                lat = np.array(lat) * 1e3
                raw = blob.cpu().numpy()
                lab = raw[:B * args.k * 8].view(np.int64).reshape(B, args.k)
                hlat = []
                for _ in range(min(steps, 50)):
                    t1 = time.perf_counter()
                    hl, hd, hf = idx.search_batch(qs, args.k)
                    hlat.append(time.perf_counter() - t1)
                out[f"batch{B}"] = {"queries_per_s": steps * B / el, "ms_per_step": el / steps * 1e3, "steps": steps,
                                    "warmup": args.warmup, "p50_ms": float(np.percentile(lat, 50)),
                                    "p95_ms": float(np.percentile(lat, 95)),
                                    "host_api_p50_ms": float(np.percentile(np.array(hlat) * 1e3, 50)),
                                    "planted_top1_ok": bool(lab[0][0] == 1 + (4242 % args.rows)),
                                    "planted_labels": [int(v) for v in lab[0]],
                                    "host_api_labels_equal": bool(np.array_equal(hl[0].astype(np.int64), lab[0])),
                                    "timing": "queries_per_s: K dawn_index_search_device calls on device-resident queries between two "
                                              "device synchronisations; p50 / p95: the same call synchronised every time"}
            out["shard_info"] = idx.shard_info()
            out["stats"] = idx.stats()
        except Exception as e:  # keep what was measured; the parent reports the failure
            out["error"] = repr(e)
        print(json.dumps(out), flush=True)
    idx.close()


if __name__ == "__main__":
    main()
