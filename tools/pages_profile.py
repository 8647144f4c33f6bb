"""Pages leg of the embedder alone, for rocprofv3 --kernel-trace --stats (dev tool): 256 pages x 128 tokens, 10 passes;
with an argument N: N sequences of 8..32 tokens (the query batch of configs[2])."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth

with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    if len(sys.argv) > 1:
        seqs = synth.token_sequences(5, int(sys.argv[1]), 8, 32)
    else:
        seqs = synth.token_sequences(5, 256, 128, 128)
    for _ in range(12):
        ep.calculate_embedding(seqs)
