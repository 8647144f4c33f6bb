import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
from oracle import oracle_lib as O
for n in (127, 200, 1000):
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, n, 1)
    x = O.unit_rows(1, 0, n)
    q = synth.unit_rows(2, 0, 1)[0]
    exact = 1.0 - x.astype(np.float64) @ q.astype(np.float64)
    order = np.argsort(exact)
    lab, dist = idx.search(q, 64)
    got = set(int(l) - 1 for l in lab)
    want = set(int(r) for r in order[:64])
    missing = sorted(want - got)
    print(n, "fallbacks", idx.stats(), "missing rows", missing[:20], "their ranks", [int(np.where(order == r)[0][0]) for r in missing[:20]])
    extra = sorted(got - want)
    print("   extra rows", extra[:20], "ranks", [int(np.where(order == r)[0][0]) for r in extra[:20]])
