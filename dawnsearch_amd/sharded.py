"""Row-sharded search across the GPUs of one node: one process per GPU (`torch.distributed`, backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

This is the MI355X counterpart of the reference's only distributed step — `SearchService::search_remote`
(src/search/search_service.rs:201-277) fanning a query vector out to peer instances and merging their
top-20 lists with `BestResults` — re-designed for GPUs that share a node: every rank scans its contiguous
row range with identical queries and writes its `(label u64, distance f32)[B][k]` lists + found counts into ONE
packed blob (<= 62 KB per rank at B=256, k=20: latency-bound), the blobs are all-gathered with a single
collective, and a stable G-way merge (ties -> lower shard = earlier rows) on every rank reproduces the
single-index answer bit for bit.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

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
    @staticmethod
    def blob_layout(B: int, count: int):
        """(total bytes, offset of distances, offset of found) of one shard's result blob (dawn_hip.h)."""
        return lib.dawn_result_blob_bytes(B, count), B * count * 8, B * count * 12

    def gather_merge_blob(self, blob, B: int, count: int):
        """blob: this rank's packed results (uint8 tensor of blob_layout(B,count)[0] bytes).  ONE all-gather, then
        the merge: RCCL + dawn_topk_merge_packed_device on the current stream for device tensors, gloo + host
        merge for CPU tensors.  Returns (labels int64 [B,count], distances f32 [B,count], found int32 [B])."""
        import torch
        nbytes, off_d, off_f = self.blob_layout(B, count)
        assert blob.dtype == torch.uint8 and blob.numel() == nbytes
        if self.world == 1:
            g_blob = blob
        else:
            g_blob = torch.empty((self.world * nbytes,), dtype=torch.uint8, device=blob.device)
            self.dist.all_gather_into_tensor(g_blob, blob, group=self.group)
        if blob.is_cuda:
            from .index import topk_merge_packed_device
            o_lab = torch.empty((B, count), dtype=torch.int64, device=blob.device)
            o_dist = torch.empty((B, count), dtype=torch.float32, device=blob.device)
            o_found = torch.empty((B,), dtype=torch.int32, device=blob.device)
            topk_merge_packed_device(blob.device.index, self.world, B, count, g_blob.data_ptr(), o_lab.data_ptr(),
                                     o_dist.data_ptr(), o_found.data_ptr(), torch.cuda.current_stream().cuda_stream)
            return o_lab, o_dist, o_found
        g = g_blob.numpy().reshape(self.world, nbytes)
        labs = np.stack([g[r, :off_d].view(np.uint64).reshape(B, count) for r in range(self.world)])
        dsts = np.stack([g[r, off_d:off_f].view(np.float32).reshape(B, count) for r in range(self.world)])
        fnds = np.stack([g[r, off_f:off_f + 4 * B].view(np.uint32) for r in range(self.world)])
        ol, od, of = merge_host(labs, dsts, fnds, count)
        return (torch.from_numpy(ol.view(np.int64)), torch.from_numpy(od), torch.from_numpy(of.view(np.int32)))

    def gather_merge(self, labels, distances, found, count: int):
        """Shard-local results as separate torch tensors ([B,count] int64, [B,count] float32, [B] int32): packed
        into one blob, then `gather_merge_blob`."""
        import torch
        B = labels.shape[0]
        nbytes, off_d, off_f = self.blob_layout(B, count)
        blob = torch.zeros((nbytes,), dtype=torch.uint8, device=labels.device)
        blob[:off_d] = labels.contiguous().view(torch.uint8).reshape(-1)
        blob[off_d:off_f] = distances.contiguous().view(torch.uint8).reshape(-1)
        blob[off_f:off_f + 4 * B] = found.contiguous().view(torch.uint8).reshape(-1)
        return self.gather_merge_blob(blob, B, count)

    # -- full search on device-resident queries ----------------------------------------------------
    def search_device(self, d_queries, count: int):
        """d_queries: cuda float32 [B,384] (identical on every rank).  The scan writes straight into the packed
        blob; returns merged (labels, distances, found)."""
        import torch
        B = d_queries.shape[0]
        nbytes, off_d, off_f = self.blob_layout(B, count)
        blob = torch.empty((nbytes,), dtype=torch.uint8, device=d_queries.device)
        p = blob.data_ptr()
        self.index.search_device(d_queries.data_ptr(), B, count, p, p + off_d, p + off_f,
                                 torch.cuda.current_stream().cuda_stream)
        return self.gather_merge_blob(blob, B, count)
