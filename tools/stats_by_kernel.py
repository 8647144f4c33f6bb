"""Per-kernel totals of a rocprofv3 rocpd database: python tools/stats_by_kernel.py results.db [name filter]"""
import re
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
rows = [dict(zip(cols, r)) for r in c.execute("select * from kernels order by start")]
agg = defaultdict(lambda: [0, 0])
for r in rows:
    n = r["name"]
    m = re.search(r"dawn::([\w<>, ]+?)\(", n) or re.search(r"_ZN4dawn\d+(\w+)", n)
    key = (m.group(1) if m else n[:40]) + f" g={r.get('grid_size', r.get('grid_x', '?'))}"
    agg[key][0] += 1
    agg[key][1] += r["end"] - r["start"]
tot = sum(v[1] for v in agg.values())
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[:70]:70s} calls {n:6d}  total {t/1e3:10.1f} us  avg {t/n/1e3:8.2f} us  {100*t/tot:5.1f} %")
