#!/usr/bin/env python3
"""Index file I/O rates: save / load of a >= 10 GB packed index file and load of a PageEntry .emb file,
through the C ABI.  usage: loader_bench.py [rows] [dir]   (default 8 M rows = 12.4 GB in /tmp)"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
base = sys.argv[2] if len(sys.argv) > 2 else tempfile.gettempdir()
d = tempfile.mkdtemp(prefix="dawn_io_", dir=base)
p = os.path.join(d, "index.dawn")
out = {"rows": rows, "file_GB": (24 + rows * 1544) / 1e9}
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
try:
    t0 = time.perf_counter(); idx.save(p); out["save_GBps"] = out["file_GB"] / (time.perf_counter() - t0)
    for name, kw in (("load_GBps", {}), ("load_sharded4_GBps", {"devices": [0, 0, 0, 0]}), ("load_bf16_GBps", {"dtype": "bf16"})):
        o = dawn.VectorIndex(0, **kw) if "devices" not in kw else dawn.VectorIndex(**kw)
        t0 = time.perf_counter(); o.load(p); out[name] = out["file_GB"] / (time.perf_counter() - t0)
        assert o.size() == rows
        o.close()
    os.remove(p)
    # PageEntry records (1568 B, vector at byte 16): written in slabs from the index's own rows
    n_pe = min(rows, 2_000_000)
    pe = os.path.join(d, "x.warc.emb")
    with open(pe, "wb") as f:
        for o0 in range(0, n_pe, 250_000):
            m = min(250_000, n_pe - o0)
            r, _ = idx.get_rows(o0, m)
            rec = np.zeros((m, 1568), dtype=np.uint8)
            rec[:, 16:16 + 1536] = r.view(np.uint8).reshape(m, 1536)
            f.write(rec.tobytes())
    o = dawn.VectorIndex(0)
    t0 = time.perf_counter(); o.load_page_entries(pe, 1); dt = time.perf_counter() - t0
    out["page_entries"] = {"records": n_pe, "file_GB": n_pe * 1568 / 1e9, "load_GBps": n_pe * 1568 / 1e9 / dt}
    assert o.size() == n_pe
    o.close()
    os.remove(pe)
finally:
    for fn in os.listdir(d):
        os.remove(os.path.join(d, fn))
    os.rmdir(d)
print(json.dumps(out))
