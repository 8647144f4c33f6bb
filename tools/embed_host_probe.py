"""The embedder's synchronous host call (token ids in host memory -> vectors in host memory: what calculate_embedding is behind the
tokenizer) against the forward alone on the device, one text and 256 texts (dev tool).   python tools/embed_host_probe.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402


def p50(fn, calls=300):
    ts = []
    for i in range(calls + 30):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.percentile(np.array(ts[30:]) * 1e3, 50))


with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for name, seqs in (("1 text of 27 tokens", synth.token_sequences(5, 1, 27, 27)), ("1 text of 12 tokens", synth.token_sequences(5, 1, 12, 12)),
                       ("256 texts of 4-32 tokens", synth.token_sequences(5, 256, 4, 32)), ("1 page of 128 tokens", synth.token_sequences(5, 1, 128, 128))):
        ids = np.concatenate(seqs).astype(np.uint32)
        offs = np.zeros(len(seqs) + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
        d_offs = torch.from_numpy(offs).to(dev)
        out = torch.zeros((len(seqs), 384), dtype=torch.float32, device=dev)
        mx = int(max(len(s) for s in seqs))

        def dev_call():
            ep.forward_device(d_ids.data_ptr(), d_offs.data_ptr(), len(seqs), int(offs[-1]), mx, out.data_ptr(), stream)
            torch.cuda.synchronize()

        ep.set_option("host_io", 0)
        t_host0 = p50(lambda: ep.calculate_embedding(seqs))
        ep.set_option("host_io", 1)
        t_host = p50(lambda: ep.calculate_embedding(seqs))
        t_dev = p50(dev_call)
        # back-to-back on the stream: the forward's own time without the per-call synchronisation
        n = 50
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            ep.forward_device(d_ids.data_ptr(), d_offs.data_ptr(), len(seqs), int(offs[-1]), mx, out.data_ptr(), stream)
        torch.cuda.synchronize()
        t_b2b = (time.perf_counter() - t0) / n * 1e3
        print(f"{name}: host call (ids and vectors in host memory) {t_host:.4f} ms (three copy commands: {t_host0:.4f}); device-resident call + sync {t_dev:.4f} ms; "
              f"back to back {t_b2b:.4f} ms", flush=True)
