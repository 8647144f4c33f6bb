"""Dev tool: isolate a shadow-filter mismatch (conversion + DMA kernel via debug_filter_scores, stream kernel via search)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dawnsearch_amd as dawn
from dawnsearch_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 127
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, n, 1)
x = synth.unit_rows(1, 0, n)
Q = synth.unit_rows(2, 0, 3)
fs = idx.debug_filter_scores(Q)
ref = Q @ x.T
print("dense filter max err", np.abs(fs - ref[:, :fs.shape[1]]).max())
for q in Q:
    lab, dist = idx.search(q, 5)
    order = np.argsort(-(x @ q), kind="stable")[:5] + 1
    print("stream", lab, "ref", order, idx.stats())
idx.set_option("f16_shadow_b1", 0)
for q in Q:
    lab, dist = idx.search(q, 5)
    print("f32   ", lab)
idx.set_option("f16_shadow_b1", 1)
for q in Q[:2]:
    sc, rows = idx.debug_stream_lists(q)
    fs = (x @ q)
    blk = [(b, int((rows[b] != 0xFFFFFFFF).sum())) for b in range(len(rows)) if (rows[b] != 0xFFFFFFFF).any()]
    print("non-empty lists", blk)
    b0 = blk[0][0]
    got = rows[b0][rows[b0] != 0xFFFFFFFF]
    print("rows in list", np.sort(got))
    print("missing", sorted(set(range(n)) - set(got.tolist()))[:80])
    err = np.abs(sc[b0][:len(got)] - fs[got])
    print("score err max", err.max(), "sorted desc", bool(np.all(np.diff(sc[b0][:len(got)]) <= 0)))
