"""Page forward against the oracle (dev tool; the oracle is the checker here, as in tests/): max error of the unit vectors of a
batch of 65..128-token pages (bf16x3 dense layers, matrix-core attention) and of one page alone, and the time of 256 pages."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
from oracle import oracle_lib
with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    sb = oracle_lib.SynthBert(3)
    seqs = synth.token_sequences(7, 40, 65, 128)
    emb = ep.calculate_embedding(seqs)
    err = max(np.abs(emb[i] - sb.embed(seqs[i])).max() for i in range(0, 40, 3))
    print("pages batch (bf16x3 path) max err vs oracle:", err)
    one = ep.calculate_embedding([seqs[0]])
    print("single page err:", np.abs(one[0] - sb.embed(seqs[0])).max())
    seqs = synth.token_sequences(5, 256, 128, 128)
    for _ in range(3): ep.calculate_embedding(seqs)
    t0 = time.perf_counter()
    for _ in range(10): ep.calculate_embedding(seqs)
    print("256 pages ms:", (time.perf_counter() - t0) / 10 * 1e3)
