"""Dev tool: one configuration of tests/test_stress_gpu.py outside pytest (stderr visible).  python tools/stress_repro.py <seed>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_stress_gpu.py")).read()
ns = {}
exec("import numpy as np\n" + src[src.index("def _case(rng):"):src.index('@pytest.mark.parametrize("seed"')], ns)
seed = int(sys.argv[1])
n, dist, dtype, k, B, opts = ns["_case"](np.random.default_rng(1000 + seed))
for a in sys.argv[2:]:
    name, v = a.split("=")
    if name == "k":
        k = int(v)
    elif name == "B":
        B = int(v)
    else:
        opts[name] = int(v)
print("case", n, dist, dtype, k, B, opts, flush=True)
idx = dawn.VectorIndex(0, dtype=dtype)
if dist:
    idx.set_option("synth_dist", dist)
for name, v in opts.items():
    idx.set_option(name, v)
idx.fill_synthetic(1, 0, n, 1)
QROW0 = 1 << 40
if dist:
    Q = np.concatenate([synth.unit_rows_topical(1, QROW0 + 256 * i, 1, runs=(dist == 5)) for i in range(B)])
else:
    Q = synth.unit_rows(2 + seed, 0, B)
print("searching", flush=True)
out = idx.search_batch(Q, k) if B > 1 else idx.search(Q[0], k)
print("done", idx.stats(), flush=True)
