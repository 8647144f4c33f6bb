"""Per-kernel PMC sums of a rocprofv3 --pmc run (rocpd sqlite): python tools/pmc_summary.py results.db [min_duration_us]
Prints, per kernel and counter, the mean over dispatches longer than min_duration_us, plus the mean duration."""
import re, sqlite3, sys
db = sys.argv[1]
mind = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
c = sqlite3.connect(db)
kc = [d[1] for d in c.execute("pragma table_info('kernels')")]
ks = {}
for r in c.execute("select * from kernels"):
    r = dict(zip(kc, r))
    ks[r["dispatch_id"]] = (r["name"], (r["end"] - r["start"]) / 1e3)
pc = [d[1] for d in c.execute("pragma table_info('pmc_events')")]
acc = {}
for r in c.execute("select * from pmc_events"):
    r = dict(zip(pc, r))
    name, dur = ks.get(r["dispatch_id"], ("?", 0))
    if dur < mind:
        continue
    m = re.search(r"(scan_\w+?kernel|\w+_kernel)", name)
    key = ((m.group(1) if m else name[:40]) + (re.search(r"<[^>]*>", name).group(0) if "<" in name else ""), r["counter_name"])
    e = acc.setdefault(key, [0, 0.0, 0.0])
    e[0] += 1
    e[1] += r["counter_value"]
    e[2] += dur
for (k, cn), (n, v, d) in sorted(acc.items()):
    print(f"{k:28s} {cn:28s} n={n:3d} mean={v / n:16.1f} mean_dur_us={d / n:10.1f}")
