"""One-text forward A/B of embedder options (dev tool): device-resident loop, graph replay.  python tools/embed_ab.py"""
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
for rnd in range(2):
    for L in (8, 12, 20, 27, 32):
        seqs = synth.token_sequences(5, 1, L, L)
        d_ids = torch.from_numpy(np.concatenate(seqs).astype(np.int32)).to(dev)
        d_off = torch.from_numpy(np.array([0, L], dtype=np.int32)).to(dev)
        d_out = torch.zeros((1, 384), dtype=torch.float32, device=dev)
        row = []
        for name, opts in (("round 4", {"attention_wave": 2, "ffn2_split": 0, "fused_embed": 0}),
                           ("register attention", {"attention_wave": 0, "ffn2_split": 0, "fused_embed": 0}),
                           ("+ FFN-down split 4", {"attention_wave": 0, "ffn2_split": 4, "fused_embed": 0}),
                           ("+ embeddings in Q|K|V", {"attention_wave": 0, "ffn2_split": 4, "fused_embed": 1})):
            for o, v in opts.items():
                ep.set_option(o, v)
            for _ in range(10):
                ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), 1, L, L, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), 1, L, L, d_out.data_ptr(), stream)
            torch.cuda.synchronize()
            row.append(f"{name} {(time.perf_counter() - t0) / 300 * 1e3:.4f}")
        print(f"len {L:3d}: " + "   ".join(row) + " ms per forward", flush=True)
