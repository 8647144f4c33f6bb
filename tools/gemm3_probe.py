"""bf16x3 GEMM probe: accuracy against float64 and time against the f32-MFMA tile kernel (dev tool)."""
import ctypes as C
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth, _lib

with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
w = synth.bert_weights(3)
rng = np.random.default_rng(1)
for T in (37, 200, 2500):
    x = (rng.standard_normal((T, 384)) * np.linspace(0.05, 4, T)[:, None]).astype(np.float32)
    z = x.astype(np.float64) @ w["encoder.layer.0.intermediate.dense.weight"].astype(np.float64).T + w["encoder.layer.0.intermediate.dense.bias"]
    want = 0.5 * z * (1 + np.tanh(np.sqrt(2 / np.pi) * z * (1 + 0.044715 * z * z)))
    for op in (3, 4, 5):
        got = ep.debug_op(op, x, T, out_cols=1536)
        err = np.abs(got - want)
        print(f"T={T} op={op}: max abs err {err.max():.3e}  (|want| max {np.abs(want).max():.2f})  rel {np.max(err / (np.abs(want) + 1e-3)):.3e}", flush=True)
ms = C.c_double(0)
names = {0: "f32 MFMA", 1: "bf16x3 -> f32", 2: "bf16x3 -> planes", 3: "bf16x3 -> gelu planes"}
for big, var, pers in ((1 << 30, 0, 0), (0, 0, 0), (0, 1, 0), (0, 1, 256)):
    ep.set_option("gemm3_big_min_tiles", big)
    ep.set_option("gemm3_pingpong", var)
    ep.set_option("gemm3_persistent", pers)
    print(f"128x128 kernel from tiles {big}, ping-pong {var}, persistent blocks {pers}")
    if big == 0 and var < 2:
        for T in (200, 2500):  # (accuracy of this form; 2500 rows: a ragged last tile)
            x = (rng.standard_normal((T, 384)) * np.linspace(0.05, 4, T)[:, None]).astype(np.float32)
            z = x.astype(np.float64) @ w["encoder.layer.0.intermediate.dense.weight"].astype(np.float64).T + w["encoder.layer.0.intermediate.dense.bias"]
            want = 0.5 * z * (1 + np.tanh(np.sqrt(2 / np.pi) * z * (1 + 0.044715 * z * z)))
            for op in (4, 5):
                got = ep.debug_op(op, x, T, out_cols=1536)
                print(f"  T={T} op={op}: max abs err {np.abs(got - want).max():.3e}", flush=True)
    for T in (4708, 32768):
        for (N, K) in ((1152, 384), (384, 384), (1536, 384), (384, 1536)):
            r = {}
            for variant in ((0, 1, 2, 3) if var == 0 else (1, 2, 3)):
                if variant == 3 and N != 1536:
                    continue
                _lib.check(_lib.lib.dawn_embedder_debug_gemm_time(ep._h, T, N, K, variant, 20, C.byref(ms)))
                r[variant] = ms.value
            fl = 2.0 * T * N * K
            print(f"T={T:6d} N={N:5d} K={K:5d}: " + "   ".join(f"{names[v]} {t*1e3:7.1f} us ({fl/t/1e9:5.1f} TF)" for v, t in r.items()), flush=True)
