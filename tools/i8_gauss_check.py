"""Dev tool: the int8 filters on Gaussian unit rows (heavier tails than the synthetic uniform rows): how often do the
64-row and the 1024-row certificates hold?  python tools/i8_gauss_check.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(7)
idx = dawn.VectorIndex(0)
done = 0
while done < rows:
    n = min(250_000, rows - done)
    x = rng.standard_normal((n, 384), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    idx.add_batch(np.arange(done + 1, done + n + 1, dtype=np.uint64), x)
    done += n
Q = rng.standard_normal((256, 384), dtype=np.float32)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
for k in (10, 20):
    for B in (1, 64, 256):
        res = {}
        for i8 in (0, 1):
            idx.set_option("i8_shadow", i8)
            s0 = idx.stats()
            outs = [idx.search_batch(Q[j:j + B], k) for j in range(0, 256, B)]
            s1 = idx.stats()
            t0 = time.time()
            for j in range(0, 256, B):
                idx.search_batch(Q[j:j + B], k)
            ms = (time.time() - t0) / (256 / B) * 1e3
            res[i8] = (np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]), ms,
                       s1["second_chances"] - s0["second_chances"], s1["fallbacks"] - s0["fallbacks"])
        same = bool(np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]))
        print(f"gaussian rows={rows} k={k} B={B:3d}: f16 {res[0][2]:7.3f} ms (2nd {res[0][3]}, exact {res[0][4]})  "
              f"i8 {res[1][2]:7.3f} ms (2nd {res[1][3]}, exact {res[1][4]}) of 256 queries  identical={same}", flush=True)
