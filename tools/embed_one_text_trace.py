"""One text through the embedder under rocprofv3 --kernel-trace (dev tool): the device-resident forward of a 27-token text, 50 times.
rocprofv3 --kernel-trace -d DIR -o t -- python3 tools/embed_one_text_trace.py ; then  python3 tools/embed_one_text_trace.py DIR/..._results.db"""
import os
import sys

if len(sys.argv) > 1 and sys.argv[1].endswith(".db"):
    import re
    import sqlite3
    c = sqlite3.connect(sys.argv[1])
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    rows = [dict(zip(cols, r)) for r in c.execute("select * from kernels order by start")]
    def short(n):
        m = re.search(r"dawn::(\w+(<[^>]*>)?)", n)
        return m.group(1) if m else n[:40]
    seq = [(short(r["name"]), r["start"], r["end"]) for r in rows]
    # a forward starts with BertEmbeddings: a launch of its own, or (option "fused_embed", the default) the PARTS = 0 form of the
    # first layer's Q|K|V launch; it ends with the pooling launch
    starts = [i for i, s in enumerate(seq) if s[0].startswith("embed_ln") or s[0].startswith("gemm_skinny16_ln_kernel<0, 0>")]
    if len(starts) < 3:
        ends = [i for i, s in enumerate(seq) if "pool_norm" in s[0]]
        starts = [i + 1 for i in ends[:-1]]
    i0, i1 = starts[-3], starts[-2]  # one whole forward near the end
    tot = (seq[i1][1] - seq[i0][1]) / 1e3
    print(f"one forward: {i1 - i0} launches, {tot:.1f} us start to start")
    busy = 0.0
    for j in range(i0, i1):
        n, s, e = seq[j]
        gap = (seq[j + 1][1] - e) / 1e3
        busy += (e - s) / 1e3
        print(f"{n:44s} dur {((e - s) / 1e3):6.2f} us   gap after {gap:5.2f} us")
    print(f"sum of kernel durations {busy:.1f} us, gaps {tot - busy:.1f} us")
    sys.exit(0)

import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as d:
    st, cj = dawn.write_synthetic_model(d, seed=3)
    ep = dawn.EmbeddingProvider(st, cj, 0)
stream = torch.cuda.current_stream().cuda_stream
L = int(os.environ.get("LEN", "27"))
seqs = synth.token_sequences(5, 1, L, L)
d_ids = torch.from_numpy(np.concatenate(seqs).astype(np.int32)).to(dev)
d_off = torch.from_numpy(np.array([0, L], dtype=np.int32)).to(dev)
d_out = torch.zeros((1, 384), dtype=torch.float32, device=dev)
for _ in range(50):
    ep.forward_device(d_ids.data_ptr(), d_off.data_ptr(), 1, L, L, d_out.data_ptr(), stream)
torch.cuda.synchronize()
