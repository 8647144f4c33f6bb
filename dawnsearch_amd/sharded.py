"""Row-sharded search across the GPUs of one node: one process per GPU (`torch.distributed`, backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

This is the MI355X counterpart of the reference's only distributed step — `SearchService::search_remote`
(src/search/search_service.rs:201-277) fanning a query vector out to peer instances and merging their
top-20 lists with `BestResults` — re-designed for GPUs that share a node: every rank scans its contiguous
row range with identical queries, the per-shard `(label u64, distance f32)[B][k]` lists (<= 61 KB per rank at
B=256, k=20: latency-bound, one collective per array) are all-gathered, and a stable G-way merge (ties ->
lower shard = earlier rows) on every rank reproduces the single-index answer bit for bit.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import check, lib


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range [first, first+n) of `rank`: ceil-div blocks, last ranks may be short/empty."""
    per = -(-total_rows // world)
    first = min(rank * per, total_rows)
    return first, min(per, total_rows - first)


def merge_host(labels: np.ndarray, distances: np.ndarray, found: np.ndarray, count: int):
    """labels/distances [G,B,count], found [G,B] -> merged ([B,count], [B,count], [B]) on the host."""
    G, B = found.shape
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    distances = np.ascontiguousarray(distances, dtype=np.float32)
    found = np.ascontiguousarray(found, dtype=np.uint32)
    ol = np.zeros((B, count), dtype=np.uint64)
    od = np.zeros((B, count), dtype=np.float32)
    of = np.zeros(B, dtype=np.uint32)
    p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    check(lib.dawn_topk_merge_host(G, B, count, p(labels), p(distances), p(found), p(ol), p(od), p(of)))
    return ol, od, of


class ShardedSearch:
    """Per-rank handle: `index` is this rank's VectorIndex (None on a CPU-only rank in the gloo tests, where the
    shard-local lists are supplied by the caller)."""

    def __init__(self, index=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.index = index

    # -- the exchange step ------------------------------------------------------------------------
    def gather_merge(self, labels, distances, found, count: int):
        """All-gather shard-local results (torch tensors [B,count],[B,count],[B]; int64/float32/int32) and merge.
        Device tensors -> RCCL + dawn_topk_merge_device on the current stream; CPU tensors -> gloo + host merge."""
        import torch
        B = labels.shape[0]
        if self.world == 1:
            return labels, distances, found
        # concatenation layout ([world*B, count]) == stacked [world][B][count] in memory; gloo insists on it
        g_lab = torch.empty((self.world * B, count), dtype=labels.dtype, device=labels.device)
        g_dist = torch.empty((self.world * B, count), dtype=distances.dtype, device=labels.device)
        g_found = torch.empty((self.world * B,), dtype=found.dtype, device=labels.device)
        self.dist.all_gather_into_tensor(g_lab, labels.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(g_dist, distances.contiguous(), group=self.group)
        self.dist.all_gather_into_tensor(g_found, found.contiguous(), group=self.group)
        g_lab = g_lab.view(self.world, B, count)
        g_dist = g_dist.view(self.world, B, count)
        g_found = g_found.view(self.world, B)
        if labels.is_cuda:
            from .index import topk_merge_device
            o_lab = torch.empty_like(labels)
            o_dist = torch.empty_like(distances)
            o_found = torch.empty_like(found)
            topk_merge_device(labels.device.index, self.world, B, count, g_lab.data_ptr(), g_dist.data_ptr(),
                              g_found.data_ptr(), o_lab.data_ptr(), o_dist.data_ptr(), o_found.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
            return o_lab, o_dist, o_found
        ol, od, of = merge_host(g_lab.numpy().view(np.uint64), g_dist.numpy(), g_found.numpy().view(np.uint32), count)
        return (torch.from_numpy(ol.view(np.int64)), torch.from_numpy(od), torch.from_numpy(of.view(np.int32)))

    # -- full search on device-resident queries ----------------------------------------------------
    def search_device(self, d_queries, count: int):
        """d_queries: cuda float32 [B,384] (identical on every rank).  Returns merged (labels, distances, found)."""
        import torch
        B = d_queries.shape[0]
        dev = d_queries.device
        lab = torch.empty((B, count), dtype=torch.int64, device=dev)
        dist_ = torch.empty((B, count), dtype=torch.float32, device=dev)
        found = torch.empty((B,), dtype=torch.int32, device=dev)
        self.index.search_device(d_queries.data_ptr(), B, count, lab.data_ptr(), dist_.data_ptr(), found.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
        return self.gather_merge(lab, dist_, found, count)
