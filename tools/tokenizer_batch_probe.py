"""dawn_tokenizer_encode_batch per batch size: the C call alone (dev tool; run it where the cores are: this container serialises threads).
python tools/tokenizer_batch_probe.py"""
import ctypes as C
import importlib
import os
import pathlib
import sys
import tempfile
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd._lib import lib  # noqa: E402

tc = importlib.import_module("test_configs_gpu")
print("cores:", os.cpu_count())
with tempfile.TemporaryDirectory() as d:
    vocab, stems, syll = tc._vocab_file(pathlib.Path(d))
    tk = dawn.Tokenizer(vocab)
    for B in (1, 16, 31, 32, 64, 128, 256, 1000):
        texts = tc._texts(11, B, stems, syll, 2, 22)
        raws = [t.encode("utf-8") for t in texts]
        arr = (C.c_char_p * len(raws))(*raws)
        cap = sum(4 * len(r) + 8 for r in raws) + 8
        out = np.zeros(cap, dtype=np.uint32)
        offs = np.zeros(len(raws) + 1, dtype=np.int32)
        ts = []
        for i in range(80):
            t0 = time.perf_counter()
            lib.dawn_tokenizer_encode_batch(tk._h, arr, len(raws), C.c_void_p(out.ctypes.data), cap, C.c_void_p(offs.ctypes.data))
            ts.append(time.perf_counter() - t0)
        flat, o2 = tk.encode_batch(texts)
        same = all(np.array_equal(flat[o2[b]:o2[b + 1]], tk.encode(texts[b])) for b in range(0, B, max(1, B // 16)))
        print(f"B={B:4d}: {np.percentile(ts[10:], 50) * 1e6:8.1f} us per call ({np.percentile(ts[10:], 50) * 1e6 / B:5.1f} us per text), {offs[-1]} tokens; "
              f"batch = one by one: {same}", flush=True)
