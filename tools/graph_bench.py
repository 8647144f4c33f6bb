"""Dev tool: one text through the embedder (host API), plain launches vs hipGraph replay: python tools/graph_bench.py"""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    for L in (8, 27, 64, 128):
        for B in (1, 4):
            seqs = synth.token_sequences(5, B, L, L)
            row = []
            for g in (0, 1):
                ep.set_option("graphs", g)
                for _ in range(5):
                    ep.calculate_embedding(seqs)
                t0 = time.time()
                for _ in range(200):
                    ep.calculate_embedding(seqs)
                row.append((time.time() - t0) / 200 * 1e3)
            print(f"B={B} len={L:3d}: plain {row[0]:.3f} ms   graph replay {row[1]:.3f} ms", flush=True)
