"""One-text embedder latency: host API and device-resident loop, hipGraph replay on / off (dev tool)."""
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
for L in (12, 27, 128):
    seqs = synth.token_sequences(5, 1, L, L)
    d_ids = torch.from_numpy(np.concatenate(seqs).astype(np.int32)).to(dev)
    d_off = torch.from_numpy(np.array([0, L], dtype=np.int32)).to(dev)
    d_out = torch.zeros((1, 384), dtype=torch.float32, device=dev)
    for graphs, wave in ((1, 1), (1, 0), (0, 1)):
        ep.set_option("graphs", graphs)
        ep.set_option("attention_wave", wave)
        for _ in range(5):
            ep.calculate_embedding(seqs)
        t0 = time.perf_counter()
        for _ in range(200):
            ep.calculate_embedding(seqs)
        host = (time.perf_counter() - t0) / 200 * 1e3
        for _ in range(5):
            ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), 1, L, L, d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), 1, L, L, d_out.data_ptr(), stream)
        torch.cuda.synchronize()
        devl = (time.perf_counter() - t0) / 200 * 1e3
        print(f"len {L:4d} graphs={graphs} attention_wave={wave}: host API {host:.3f} ms/call   device loop {devl:.3f} ms/forward", flush=True)
