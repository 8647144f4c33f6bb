"""Why the f32-row stream lost 3 % between rounds 2 and 3 (dev tool): the same launch before and after other buffers come and go.
python tools/stream_alloc_probe.py [rows=100000000]   (run with DAWN_I6_SHADOW=0: the index starts without its packed shadow)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dawnsearch_amd as dawn  # noqa: E402
from dawnsearch_amd import synth  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
idx = dawn.VectorIndex(0)
idx.fill_synthetic(1, 0, rows, 1)
q = torch.from_numpy(synth.unit_rows(2, 0, 1)).to(dev)
k = 10
blob = torch.zeros((dawn.result_blob_bytes(1, k),), dtype=torch.uint8, device=dev)
p = blob.data_ptr()


def leg(tag, steps=10, warm=2):
    idx.set_option("f16_shadow_b1", 0)
    idx.profile_enable(True)
    for _ in range(warm):
        idx.search_device(q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)
    torch.cuda.synchronize()
    idx.profile_read()
    for _ in range(steps):
        idx.search_device(q.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)
    torch.cuda.synchronize()
    n, ms = idx.profile_read()
    idx.profile_enable(False)
    free, tot = torch.cuda.mem_get_info()
    print(f"{tag:70s}: f32 stream {ms / n:7.3f} ms = {rows * 1536 / (ms / n * 1e-3) / 8e12:.4f} of HBM; free {free / 1e9:.1f} GB; index memory {idx.memory()}", flush=True)
    idx.set_option("f16_shadow_b1", 1)


leg("fresh index (rows + int8 shadow)")
leg("again")
big = torch.empty((24_000_000_000,), dtype=torch.uint8, device=dev)
leg("a 24 GB torch buffer allocated, untouched")
big.fill_(1)
torch.cuda.synchronize()
leg("... written")
del big
torch.cuda.empty_cache()
leg("... freed")
idx.set_option("i6_shadow", 1)
torch.cuda.synchronize()
leg("packed shadow built (and released again by f16_shadow_b1 = 0)")
idx.set_option("i6_shadow", 0)
leg("i6_shadow = 0")
qq = torch.from_numpy(synth.unit_rows(2, 0, 1)).to(dev)
for _ in range(5):
    idx.search_device(qq.data_ptr(), 1, k, p, p + k * 8, p + k * 12, stream)  # int8 stream searches
torch.cuda.synchronize()
leg("after five int8-stream searches")
# does it recover?  (a background wipe of freed VRAM would end)
big = torch.empty((24_000_000_000,), dtype=torch.uint8, device=dev)
del big
torch.cuda.empty_cache()
t0 = time.time()
for i in range(12):
    leg(f"{time.time() - t0:5.1f} s after another 24 GB alloc + free", steps=6, warm=1)
    time.sleep(1.5)
# a small free
small = torch.empty((400_000_000,), dtype=torch.uint8, device=dev)
del small
torch.cuda.empty_cache()
leg("after a 0.4 GB alloc + free")
