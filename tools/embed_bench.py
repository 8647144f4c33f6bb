"""Embedder timing on the GPU box (dev tool): python tools/embed_bench.py"""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn
from dawnsearch_amd import synth

def flops(seqs):
    T = sum(len(s) for s in seqs)
    return sum(len(s) * (21.23e6 + 9216.0 * len(s)) for s in seqs), T

with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
    if len(sys.argv) > 1 and sys.argv[1] == "skinny":
        for L in (12, 27, 48, 64, 96, 128, 192, 256, 512):
            seqs = synth.token_sequences(5, 1, L, L)
            row = []
            for thr in (0, 16, 32, 64, 128, 256, 512):
                ep.set_option("skinny_max_rows", thr)
                ep.calculate_embedding(seqs)
                t0 = time.time()
                for _ in range(30):
                    ep.calculate_embedding(seqs)
                row.append((time.time() - t0) / 30 * 1e3)
            print(f"len {L:4d}: " + "  ".join(f"thr{t}={v:.3f}" for t, v in zip((0, 16, 32, 64, 128, 256, 512), row)), flush=True)
        for Bq in (6, 8, 12, 16, 24, 32):
            seqs = synth.token_sequences(5, Bq, 128, 128)
            row = []
            for thr in (0, 4096):
                ep.set_option("skinny_max_rows", thr)
                ep.calculate_embedding(seqs)
                t0 = time.time()
                for _ in range(20):
                    ep.calculate_embedding(seqs)
                row.append((time.time() - t0) / 20 * 1e3)
            print(f"tokens {Bq*128:5d}: tile={row[0]:.3f} ms  skinny={row[1]:.3f} ms", flush=True)
        sys.exit(0)
    cases = {
        "B=1 len 12": synth.token_sequences(5, 1, 12, 12),
        "B=1 len 27": synth.token_sequences(5, 1, 27, 27),
        "B=1 len 128": synth.token_sequences(5, 1, 128, 128),
        "B=256 len 4..32": synth.token_sequences(5, 256, 4, 32),
        "B=64 len 128": synth.token_sequences(6, 64, 128, 128),
        "B=256 len 128": synth.token_sequences(7, 256, 128, 128),
    }
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, seqs in cases.items():
        if only and only not in name:
            continue
        ep.calculate_embedding(seqs)
        it = 20
        t0 = time.time()
        for _ in range(it):
            ep.calculate_embedding(seqs)
        dt = (time.time() - t0) / it
        fl, T = flops(seqs)
        print(f"{name:18s} tokens={T:6d}  {dt*1e3:8.3f} ms/call  {len(seqs)/dt:9.0f} seq/s  {fl/dt/1e12:6.2f} TFLOP/s", flush=True)
