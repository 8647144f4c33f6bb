"""Summaries of a rocprofv3 run of bench.py (rocpd sqlite output) for profiles/:

  python tools/profile_summary.py stats <results.db> <out_dir>   -> bench_kernel_stats.csv, bench_kernel_trace_by_leg.json
  python tools/profile_summary.py pmc   <results.db> <out_dir>   -> bench_pmc_fetch_size.csv (+ prints traffic.json entries)

The runs: `rocprofv3 --kernel-trace --stats -d DIR -o bench -- python3 bench.py --steps 5 --warmup 2` and, separately,
`rocprofv3 --kernel-trace --pmc FETCH_SIZE -d DIR -o bench -- python3 bench.py --steps 5 --warmup 2`.
"""
import csv
import json
import os
import re
import sqlite3
import sys

OURS = re.compile(r"dawn::(\w+)")


def short(name):
    m = re.match(r"_ZN4dawn(\d+)", name)  # names rocprofv3 left mangled (kernels with _Float16 vector parameters)
    if m:
        k = int(m.group(1))
        base = name[m.end():m.end() + k]
        rest = name[m.end() + k:]
        t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        return base + ("<" + ", ".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">" if t else "")
    m = OURS.search(name)
    if not m:
        return None
    targs = re.search(r"dawn::\w+<([^>]*)>", name)
    return m.group(1) + ("<" + targs.group(1) + ">" if targs else "")


def dispatches(db):
    c = sqlite3.connect(db)
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    rows = c.execute("select * from kernels order by start").fetchall()
    return c, [dict(zip(cols, r)) for r in rows]


def main():
    mode, db, out = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(out, exist_ok=True)
    c, disp = dispatches(db)
    if mode == "stats":
        with open(os.path.join(out, "bench_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
            for r in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
                w.writerow(r)
        legs = {}
        for d in disp:
            s = short(d["name"])
            if s is None:
                continue
            dur = d["end"] - d["start"] if "end" in d else d["duration"]
            # one entry per (kernel, launch shape, decade of duration): the bench legs differ by index size
            key = f"{s} grid={d.get('grid_x')} wg={d.get('workgroup_x')} ~{10 ** len(str(int(dur // 1000))) // 10}us+"
            e = legs.setdefault(key, {"calls": 0, "total_ns": 0, "min_ns": None, "max_ns": 0})
            e["calls"] += 1
            e["total_ns"] += dur
            e["min_ns"] = dur if e["min_ns"] is None else min(e["min_ns"], dur)
            e["max_ns"] = max(e["max_ns"], dur)
        for e in legs.values():
            e["avg_us"] = round(e["total_ns"] / e["calls"] / 1e3, 2)
        with open(os.path.join(out, "bench_kernel_trace_by_leg.json"), "w") as f:
            json.dump(dict(sorted(legs.items(), key=lambda kv: -kv[1]["total_ns"])), f, indent=1)
        print("wrote", len(legs), "kernel/leg groups")
    else:
        pm = {}
        pc = [d[1] for d in c.execute("pragma table_info('pmc_events')")]
        for r in c.execute("select * from pmc_events"):
            r = dict(zip(pc, r))
            pm[r["dispatch_id"]] = (r["counter_name"], r["counter_value"])
        groups = {}
        with open(os.path.join(out, "bench_pmc_fetch_size.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["dispatch_id", "kernel", "grid", "workgroup", "duration_us", "FETCH_SIZE_KB", "hbm_read_bytes_x2_corrected"])
            for d in disp:
                s = short(d["name"])
                if s is None or d["dispatch_id"] not in pm:
                    continue
                dur = (d["end"] - d["start"]) / 1e3
                if dur < 3000:  # the 100M-row passes only
                    continue
                kb = pm[d["dispatch_id"]][1]
                w.writerow([d["dispatch_id"], s, d.get("grid_x"),
                            d.get("workgroup_x"), round(dur, 1), kb, kb * 1024 * 2])
                groups.setdefault(s, []).append(kb * 1024 * 2)
        for k, v in groups.items():
            print(k, "launches", len(v), "mean HBM read bytes/launch", sum(v) / len(v))

    if mode == "timeline":  # the kernels of the last `window_ms` of the trace, one line each (a batch's anatomy)
        window_ns = float(sys.argv[4]) * 1e6 if len(sys.argv) > 4 else 60e6
        ours = [d for d in disp if short(d["name"]) is not None]
        t_end = max(d["end"] for d in ours)
        last = [d for d in ours if d["start"] >= t_end - window_ns]
        t0 = last[0]["start"]
        with open(os.path.join(out, "timeline.txt"), "w") as f:
            for d in last:
                line = (f"{(d['start'] - t0) / 1e3:10.1f} us  +{(d['end'] - d['start']) / 1e3:9.1f} us  {short(d['name'])}  "
                        f"grid={d.get('grid_x')} wg={d.get('workgroup_x')}")
                print(line)
                f.write(line + "\n")


main()
