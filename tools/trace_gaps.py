import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
rows = [dict(zip(cols, r)) for r in c.execute("select * from kernels order by start")]
import re
def short(n):
    m = re.search(r"dawn::(\w+)", n) or re.search(r"_ZN4dawn\d+(\w+?)I", n)
    return m.group(1) if m else n[:30]
# last 40 dispatches of the B=1 phase: find sequences stream->merge->exact
seq = [(short(r["name"]), r["start"], r["end"]) for r in rows]
# print a window in the B=1 phase (find 60th occurrence of scan_filter_i8s)
idxs = [i for i, s in enumerate(seq) if "scan_filter_i8s" in s[0]]
i0 = idxs[100]
for j in range(i0, i0 + 9):
    n, s, e = seq[j]
    gap = (s - seq[j - 1][2]) / 1e3
    print(f"{n:28s} dur {((e - s) / 1e3):8.2f} us   gap before {gap:7.2f} us")
# and the B=256 phase
idxs = [i for i, s in enumerate(seq) if "prep_queries_i8" in s[0]]
i0 = idxs[-10]
for j in range(i0, i0 + 10):
    n, s, e = seq[j]
    gap = (s - seq[j - 1][2]) / 1e3
    print(f"{n:28s} dur {((e - s) / 1e3):8.2f} us   gap before {gap:7.2f} us")
