"""Multi-GPU path on CPU: world_size-2 (and 3) `gloo` processes run the sharded search's exchange step —
all-gather of per-shard top-k lists + stable merge — and must reproduce the single-index oracle answer.
Shard-local lists come from the oracle here (no GPU in this container); on the GPU box the same
`ShardedSearch.gather_merge` runs over RCCL with lists produced by the HIP scan (tests/test_sharded_gpu.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, k, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dawnsearch_amd as dawn
        from dawnsearch_amd import synth
        from oracle import oracle_lib as O
        first, n = dawn.shard_range(n_rows, world, rank)
        X = O.unit_rows(1, first, n) if n else np.zeros((0, 384), dtype=np.float32)
        ids = np.arange(1 + first, 1 + first + n, dtype=np.uint64)
        Q = np.concatenate([synth.unit_rows(2, 0, 3), synth.planted_queries(1, [5, n_rows - 1], 7)])
        if n_rows > 40:  # an exact index row as query: distance ~0 hit must come out first
            Q[2] = O.unit_rows(1, 3, 1)[0]
        B = len(Q)
        lab = np.zeros((B, k), dtype=np.uint64)
        dst = np.zeros((B, k), dtype=np.float32)
        fnd = np.zeros(B, dtype=np.uint32)
        for b in range(B):
            l, d = O.scan_topk(X, ids, Q[b], k)
            lab[b, :len(l)], dst[b, :len(l)], fnd[b] = l, d, len(l)
        ss = dawn.ShardedSearch(index=None)
        assert ss.world == world and ss.rank == rank
        ml, md, mf = ss.gather_merge(torch.from_numpy(lab.view(np.int64)), torch.from_numpy(dst),
                                     torch.from_numpy(fnd.view(np.int32)), k)
        # every rank holds the same merged answer == the oracle over the whole index
        Xall = O.unit_rows(1, 0, n_rows)
        idall = np.arange(1, n_rows + 1, dtype=np.uint64)
        ok = True
        for b in range(B):
            ol, od = O.scan_topk(Xall, idall, Q[b], k)
            f = int(mf[b])
            ok &= f == len(ol)
            ok &= np.array_equal(ml[b, :f].numpy().view(np.uint64), ol)
            ok &= np.array_equal(md[b, :f].numpy().view(np.uint32), od.view(np.uint32))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows,k", [(2, 5000, 10), (2, 7, 20), (3, 1001, 20)])
def test_gloo_gather_merge_matches_oracle(world, n_rows, k):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_rows, k, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_shard_range_and_host_merge(dawn):
    assert [dawn.shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert [dawn.shard_range(2, 4, r) for r in range(4)] == [(0, 1), (1, 1), (2, 0), (2, 0)]
    assert dawn.shard_range(100_000_000, 8, 7) == (87_500_000, 12_500_000)
    # ties -> lower shard first, then shard-local order; short lists handled
    lab = np.array([[[1, 2, 3]], [[11, 12, 13]]], dtype=np.uint64)
    d = np.array([[[0.1, 0.5, 0.9]], [[0.1, 0.5, 0.6]]], dtype=np.float32)
    f = np.array([[3], [2]], dtype=np.uint32)
    ol, od, of = dawn.merge_host(lab, d, f, 3)
    assert ol[0].tolist() == [1, 11, 2] and of[0] == 3
    f = np.array([[1], [0]], dtype=np.uint32)
    ol, od, of = dawn.merge_host(lab, d, f, 3)
    assert of[0] == 1 and ol[0, 0] == 1
