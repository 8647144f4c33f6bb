"""Which bf16x3 GEMM form for the 256-query batch (4708 tokens)?  Sweeps "gemm3_big_min_tiles" (128 x 128 tiles from which the
persistent 128 x 128 kernel is used) and "gemm3_stages" (dev tool).  python tools/embed_tile_sweep.py"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dawnsearch_amd as dawn
from dawnsearch_amd import synth

dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
stream = torch.cuda.current_stream().cuda_stream
for name, B, lo, hi in (("256 queries", 256, 4, 32), ("512 queries", 512, 4, 32), ("64 pages", 64, 128, 128), ("256 pages", 256, 128, 128)):
    seqs = synth.token_sequences(5, B, lo, hi)
    lens = np.array([len(x) for x in seqs]); offs = np.zeros(B + 1, dtype=np.int32); offs[1:] = np.cumsum(lens)
    T, mx = int(offs[-1]), int(lens.max())
    d_ids = torch.from_numpy(np.concatenate(seqs).astype(np.int32)).to(dev)
    d_off = torch.from_numpy(offs).to(dev)
    d_out = torch.zeros((B, 384), dtype=torch.float32, device=dev)
    ref = None
    for stages in (2, 3):
        for big in (512, 400, 300, 200, 100):
            ep.set_option("gemm3_stages", stages)
            ep.set_option("gemm3_big_min_tiles", big)
            for _ in range(3):
                ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), B, T, mx, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            out = d_out.cpu().numpy().copy()
            if ref is None:
                ref = out
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), B, T, mx, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            print(f"{name:12s} T={T:6d} stages={stages} big_min_tiles={big:4d}: {ms:7.3f} ms  max|d| vs first {np.abs(out - ref).max():.2e}", flush=True)
