"""Sharded search on the GPU box (1 GPU): the index split into G shards held as G separate VectorIndex objects,
per-shard HIP scan, device merge kernel — must equal the single-index answer and the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from dawnsearch_amd import synth  # noqa: E402


@pytest.mark.parametrize("G,k", [(2, 10), (8, 20), (8, 64), (5, 7)])
def test_device_merge_equals_single_index(dawn, oracle, G, k):
    import torch
    n = 40_000
    dev = torch.device("cuda", 0)
    full = dawn.VectorIndex(0)
    full.fill_synthetic(1, 0, n, 1)
    shards = []
    for g in range(G):
        first, m = dawn.shard_range(n, G, g)
        s = dawn.VectorIndex(0)
        s.fill_synthetic(1, first, m, 1 + first)
        shards.append(s)
    Q = np.concatenate([synth.unit_rows(2, 0, 5), synth.planted_queries(1, [0, n // G, n - 1], 3)])
    Q[1] = oracle.unit_rows(1, n // G - 1, 1)[0]  # exact hit on the last row of shard 0
    B = len(Q)
    dq = torch.from_numpy(Q).to(dev)
    g_lab = torch.zeros((G, B, k), dtype=torch.int64, device=dev)
    g_dist = torch.zeros((G, B, k), dtype=torch.float32, device=dev)
    g_found = torch.zeros((G, B), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for g, s in enumerate(shards):
        s.search_device(dq.data_ptr(), B, k, g_lab[g].data_ptr(), g_dist[g].data_ptr(), g_found[g].data_ptr(), st)
    o_lab = torch.zeros((B, k), dtype=torch.int64, device=dev)
    o_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
    o_found = torch.zeros((B,), dtype=torch.int32, device=dev)
    dawn.topk_merge_device(0, G, B, k, g_lab.data_ptr(), g_dist.data_ptr(), g_found.data_ptr(), o_lab.data_ptr(),
                           o_dist.data_ptr(), o_found.data_ptr(), st)
    torch.cuda.synchronize()
    lab1, dist1, found1 = full.search_batch(Q, k)
    ml = o_lab.cpu().numpy().view(np.uint64)
    md = o_dist.cpu().numpy()
    assert np.array_equal(o_found.cpu().numpy(), found1.astype(np.int32))
    assert np.array_equal(ml, lab1) and np.array_equal(md.view(np.uint32), dist1.view(np.uint32))
    # host merge agrees with the device merge
    hl, hd, hf = dawn.merge_host(g_lab.cpu().numpy().view(np.uint64), g_dist.cpu().numpy(),
                                 g_found.cpu().numpy().view(np.uint32), k)
    assert np.array_equal(hl, ml) and np.array_equal(hd, md)
    X = oracle.unit_rows(1, 0, n)
    ids = np.arange(1, n + 1, dtype=np.uint64)
    for b in range(B):
        ol, od = oracle.scan_topk(X, ids, Q[b], k, threads=4)
        assert np.array_equal(ml[b], ol) and np.array_equal(md[b], od)


def test_sharded_search_world1(dawn):
    import torch
    idx = dawn.VectorIndex(0)
    idx.fill_synthetic(1, 0, 5000, 1)
    ss = dawn.ShardedSearch(idx)
    Q = synth.unit_rows(2, 0, 4)
    lab, d, f = ss.search_device(torch.from_numpy(Q).cuda(), 10)
    torch.cuda.synchronize()
    l2, d2, f2 = idx.search_batch(Q, 10)
    assert np.array_equal(lab.cpu().numpy().view(np.uint64), l2) and np.array_equal(d.cpu().numpy(), d2)


@pytest.mark.parametrize("G,B,k", [(8, 6, 10), (4, 40, 20)])
def test_packed_blob_merge_equals_single_index(dawn, G, B, k):
    """The one-collective exchange: every shard's scan writes into its slot of the gathered blob array; the
    packed merge must equal the single-index answer (B = 40 goes through the matrix-core path)."""
    import torch
    n = 60_000
    dev = torch.device("cuda", 0)
    full = dawn.VectorIndex(0)
    full.fill_synthetic(1, 0, n, 1)
    Q = synth.unit_rows(2, 0, B)
    Q[B - 1] = synth.planted_queries(1, [n - 1], 3)[0]
    dq = torch.from_numpy(Q).to(dev)
    nbytes = dawn.result_blob_bytes(B, k)
    assert nbytes % 16 == 0 and nbytes >= B * k * 12 + B * 4
    g_blob = torch.zeros((G * nbytes,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    shards = []
    for g in range(G):
        first, m = dawn.shard_range(n, G, g)
        s = dawn.VectorIndex(0)
        s.fill_synthetic(1, first, m, 1 + first)
        shards.append(s)
        p = g_blob.data_ptr() + g * nbytes
        s.search_device(dq.data_ptr(), B, k, p, p + B * k * 8, p + B * k * 12, st)
    o_lab = torch.zeros((B, k), dtype=torch.int64, device=dev)
    o_dist = torch.zeros((B, k), dtype=torch.float32, device=dev)
    o_found = torch.zeros((B,), dtype=torch.int32, device=dev)
    dawn.topk_merge_packed_device(0, G, B, k, g_blob.data_ptr(), o_lab.data_ptr(), o_dist.data_ptr(),
                                  o_found.data_ptr(), st)
    torch.cuda.synchronize()
    lab1, dist1, found1 = full.search_batch(Q, k)
    assert np.array_equal(o_found.cpu().numpy(), found1.astype(np.int32))
    assert np.array_equal(o_lab.cpu().numpy().view(np.uint64), lab1)
    assert np.array_equal(o_dist.cpu().numpy().view(np.uint32), dist1.view(np.uint32))
    assert lab1[B - 1][0] == n
