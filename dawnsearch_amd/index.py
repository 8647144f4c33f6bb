"""Thin Python handles over the dawn_index_* / dawn_best_* / dawn_vec_* C ABI (host-side plumbing for
tests and bench.py; the Rust/C++ callers bind the same symbols directly — see INTEGRATION.md)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import EM_LEN, check, lib


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class VectorIndex:
    """usearch::ffi::Index replacement (search_provider.rs:102-284) living in HBM on one MI355X — or, with
    `devices=[...]`, dealt over several GPUs of the node behind the same calls (dawn_index_create_sharded)."""

    def __init__(self, device: int = 0, dims: int = EM_LEN, dtype: str = "f32", devices=None):
        h = C.c_void_p()
        code = {"f32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16}[dtype]
        if devices is not None:
            devs = (C.c_int * len(devices))(*devices)
            check(lib.dawn_index_create_sharded(dims, code, len(devices), devs, C.byref(h)))
            device = devices[0]
        else:
            check(lib.dawn_index_create(dims, code, device, C.byref(h)))
        self._h = h
        self.device = device
        self.dtype = dtype

    def shard_info(self):
        """-> {"n_shards", "gather" (1 RCCL all-gather, -1 RCCL selected / not initialised yet, 2 peer copies, 0 none),
        "sizes"}."""
        n = C.c_int(0)
        g = C.c_int(0)
        sizes = np.zeros(64, dtype=np.uintp)
        check(lib.dawn_index_shard_info(self._h, C.byref(n), C.byref(g), _ptr(sizes), 64))
        return {"n_shards": n.value, "gather": g.value, "sizes": sizes[:n.value].astype(np.int64).tolist()}

    def close(self):
        if getattr(self, "_h", None):
            lib.dawn_index_destroy(self._h)
            self._h = None

    __del__ = close

    # -- usearch surface ----------------------------------------------------------------------
    def reserve(self, n: int):
        check(lib.dawn_index_reserve(self._h, n))

    def size(self) -> int:
        return lib.dawn_index_size(self._h)

    def capacity(self) -> int:
        return lib.dawn_index_capacity(self._h)

    def add(self, id_: int, v: np.ndarray):
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.shape == (EM_LEN,)
        check(lib.dawn_index_add(self._h, id_, _ptr(v)))

    def add_batch(self, ids: np.ndarray, rows: np.ndarray):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        assert rows.ndim == 2 and rows.shape[1] == EM_LEN and ids.shape == (rows.shape[0],)
        check(lib.dawn_index_add_batch(self._h, rows.shape[0], _ptr(ids), _ptr(rows)))

    def search(self, q: np.ndarray, count: int):
        """-> (labels u64[found], distances f32[found]) ascending by distance."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        labels = np.zeros(count, dtype=np.uint64)
        dist = np.zeros(count, dtype=np.float32)
        found = C.c_size_t(0)
        check(lib.dawn_index_search(self._h, _ptr(q), count, _ptr(labels), _ptr(dist), C.byref(found)))
        return labels[:found.value], dist[:found.value]

    def search_limited(self, q: np.ndarray, count: int, distance_limit: float):
        """The answering side of a remote search (udp_service.rs:196-199): hits with distance < distance_limit only."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        labels = np.zeros(count, dtype=np.uint64)
        dist = np.zeros(count, dtype=np.float32)
        found = C.c_size_t(0)
        check(lib.dawn_index_search_limited(self._h, _ptr(q), count, C.c_float(distance_limit), _ptr(labels), _ptr(dist),
                                            C.byref(found)))
        return labels[:found.value], dist[:found.value]

    def search_batch(self, Q: np.ndarray, count: int):
        """-> (labels [B,count], distances [B,count], found [B])."""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        B = Q.shape[0]
        labels = np.zeros((B, count), dtype=np.uint64)
        dist = np.zeros((B, count), dtype=np.float32)
        found = np.zeros(B, dtype=np.uintp)
        check(lib.dawn_index_search_batch(self._h, _ptr(Q), B, count, _ptr(labels), _ptr(dist), _ptr(found)))
        return labels, dist, found.astype(np.int64)

    def save(self, path: str):
        check(lib.dawn_index_save(self._h, path.encode()))

    def load(self, path: str):
        check(lib.dawn_index_load(self._h, path.encode()))

    def load_page_entries(self, emb_path: str, first_id: int = 1):
        check(lib.dawn_index_load_page_entries(self._h, emb_path.encode(), first_id))

    # -- device-resident forms (raw device pointers as ints) ------------------------------------
    def search_device(self, d_queries: int, B: int, count: int, d_labels: int, d_distances: int, d_found: int,
                      stream: int = 0):
        check(lib.dawn_index_search_device(self._h, d_queries, B, count, d_labels, d_distances, d_found, stream))

    # -- bench / test utilities -----------------------------------------------------------------
    def fill_synthetic(self, seed: int, first_row: int, n: int, first_id: int = 1):
        check(lib.dawn_index_fill_synthetic(self._h, seed, first_row, n, first_id))

    def get_rows(self, first: int, n: int):
        rows = np.empty((n, EM_LEN), dtype=np.float32)
        ids = np.empty(n, dtype=np.uint64)
        check(lib.dawn_index_get_rows(self._h, first, n, _ptr(rows), _ptr(ids)))
        return rows, ids

    def profile_enable(self, on: bool = True):
        check(lib.dawn_index_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        n = C.c_uint64(0)
        ms = C.c_double(0.0)
        check(lib.dawn_index_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def stats(self):
        s = C.c_uint64(0)
        f = C.c_uint64(0)
        c2 = C.c_uint64(0)
        check(lib.dawn_index_stats_ext(self._h, C.byref(s), C.byref(c2), C.byref(f)))
        dp = C.c_uint64(0)
        check(lib.dawn_index_stats_deep(self._h, C.byref(dp)))
        bd, pf, dm = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib.dawn_index_stats_ladder(self._h, C.byref(bd), C.byref(pf), C.byref(dm)))
        # second_chances: the 64-row certificate failed, no exact pass needed; deepened: those settled by a deeper round;
        # bounded: every certificate failed (or the index is demoted), the bounded exact pass on the int8 shadow answered (no pass
        # over all rows); packed_failures: single queries whose packed-stream certificate failed; demoted: single queries sent to
        # the bounded pass directly by the ladder feedback
        return {"searches": s.value, "fallbacks": f.value, "second_chances": c2.value, "deepened": dp.value,
                "bounded": bd.value, "packed_failures": pf.value, "demoted": dm.value}

    def stats_raw(self):
        """The device-side counters as they are (dawn_hip_debug.h): 8 slots indexed by final flag; [5] packed-stream failures,
        [7] (row, query) pairs the bounded pass scored exactly."""
        out = (C.c_uint64 * 8)()
        check(lib.dawn_index_debug_raw_stats(self._h, out))
        return [int(v) for v in out]

    def i6_refine(self, k: int = 10):
        """(entries of its coarse list a wave of the packed stream refines for a top-k search — 0: the int8 stream is used, -1: no
        packed shadow —, the measured histogram of the shadow's error bounds in bins of 0.004): dawn_hip_debug.h."""
        n = C.c_int(0)
        frac = (C.c_float * 64)()
        check(lib.dawn_index_debug_i6_refine(self._h, k, C.byref(n), frac))
        return n.value, [float(v) for v in frac]

    def stats_batch_feedback(self):
        """Batches the FP6 first filter took / batches its feedback handed to the int8 pass / batches the int8 pass ran with the
        deeper thresholds of a ladder-heavy index (dawn_hip_debug.h)."""
        a, b, c, d = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib.dawn_index_stats_batch_feedback(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"f6_batches": a.value, "f6_suspended": b.value, "deepened_batches": c.value, "rerun_answers": d.value}

    def stats_f6(self):
        s = self.stats_batch_feedback()
        return {"f6_batches": s["f6_batches"], "f6_suspended": s["f6_suspended"]}

    def memory(self):
        """HBM bytes held by the index: rows, filter shadows built so far, everything else."""
        r = C.c_uint64(0)
        sh = C.c_uint64(0)
        o = C.c_uint64(0)
        check(lib.dawn_index_memory(self._h, C.byref(r), C.byref(sh), C.byref(o)))
        return {"rows": r.value, "shadows": sh.value, "other": o.value}

    def set_option(self, name: str, value: int):
        check(lib.dawn_index_set_option(self._h, name.encode(), value))

    def debug_filter_scores(self, queries: np.ndarray) -> np.ndarray:
        """Test hook: f16 matrix-core filter scores of the queries against rows [0, min(size, 8192))."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, EM_LEN)
        n = C.c_size_t(0)
        tmp = np.zeros((q.shape[0], min(self.size(), 8192)), dtype=np.float32)
        check(lib.dawn_index_debug_filter_scores(self._h, _ptr(q), q.shape[0], _ptr(tmp), C.byref(n)))
        return tmp[:, :n.value]


    def debug_f6_scores(self, queries: np.ndarray) -> np.ndarray:
        """Test hook: the FP6 shadow's upper-bound scores of the queries against rows [0, min(size, 8192)) (option f6_shadow)."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, EM_LEN)
        n = C.c_size_t(0)
        tmp = np.zeros((q.shape[0], min(self.size(), 8192)), dtype=np.float32)
        check(lib.dawn_index_debug_f6_scores(self._h, _ptr(q), q.shape[0], _ptr(tmp), C.byref(n)))
        return tmp[:, :n.value]

    def debug_time_full_pass(self, B: int, iters: int = 5) -> float:
        """Timing hook: mean ms of the matrix-core full pass alone (after a batched search set the thresholds)."""
        ms = C.c_double(0.0)
        check(lib.dawn_index_debug_time_full_pass(self._h, B, iters, C.byref(ms)))
        return ms.value

    def debug_stream_lists(self, query: np.ndarray, cap_blocks: int = 4096):
        """Test hook: per-workgroup candidate lists of the batch-1 streaming filter -> (scores [blocks,64], rows)."""
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(EM_LEN)
        sc = np.zeros((cap_blocks, 64), dtype=np.float32)
        rows = np.zeros((cap_blocks, 64), dtype=np.uint32)
        n = C.c_size_t(0)
        check(lib.dawn_index_debug_stream_lists(self._h, _ptr(q), _ptr(sc), _ptr(rows), cap_blocks, C.byref(n)))
        return sc[:n.value], rows[:n.value]

def _debug_stream_bound(self) -> float:
    """Test hook: the certificate bound T of the packed-shadow stream for the last debug_stream_lists query."""
    b = C.c_float(0.0)
    check(lib.dawn_index_debug_stream_bound(self._h, C.byref(b)))
    return b.value


VectorIndex.debug_stream_bound = _debug_stream_bound


def topk_merge_device(device: int, G: int, B: int, count: int, d_in_labels: int, d_in_dist: int, d_in_found: int,
                      d_labels: int, d_dist: int, d_found: int, stream: int = 0):
    check(lib.dawn_topk_merge_device(device, G, B, count, d_in_labels, d_in_dist, d_in_found, d_labels, d_dist,
                                     d_found, stream))


def result_blob_bytes(B: int, count: int) -> int:
    return lib.dawn_result_blob_bytes(B, count)


def topk_merge_packed_device(device: int, G: int, B: int, count: int, d_blobs: int, d_labels: int, d_dist: int,
                             d_found: int, stream: int = 0):
    check(lib.dawn_topk_merge_packed_device(device, G, B, count, d_blobs, d_labels, d_dist, d_found, stream))


# ---- src/search/vector.rs ---------------------------------------------------------------------

def is_normalized(v: np.ndarray) -> bool:
    v = np.ascontiguousarray(v, dtype=np.float32)
    assert v.shape == (EM_LEN,)
    return bool(lib.dawn_vec_is_normalized(_ptr(v)))


def normalize(v: np.ndarray) -> np.ndarray:
    v = np.array(v, dtype=np.float32, copy=True)
    lib.dawn_vec_normalize(_ptr(v), v.size)
    return v


def to24(v: np.ndarray) -> bytes:
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.zeros(EM_LEN * 3, dtype=np.uint8)
    lib.dawn_vec_to24(_ptr(v), _ptr(out))
    return out.tobytes()


def from24(data: bytes) -> np.ndarray:
    a = np.frombuffer(data, dtype=np.uint8).copy()
    assert a.size == EM_LEN * 3
    out = np.zeros(EM_LEN, dtype=np.float32)
    check(lib.dawn_vec_from24(_ptr(a), _ptr(out)))
    return out


class BestResults:
    """src/search/best_results.rs:28-107."""

    def __init__(self, size: int):
        h = C.c_void_p()
        check(lib.dawn_best_new(size, C.byref(h)))
        self._h = h

    def insert(self, id_: int, distance: float) -> bool:
        return bool(lib.dawn_best_insert(self._h, id_, float(np.float32(distance))))

    def sort(self):
        lib.dawn_best_sort(self._h)

    def worst_distance(self) -> float:
        return float(lib.dawn_best_worst_distance(self._h))

    def __len__(self):
        return lib.dawn_best_len(self._h)

    def results(self):
        out = []
        for i in range(len(self)):
            id_ = C.c_size_t(0)
            d = C.c_float(0)
            check(lib.dawn_best_get(self._h, i, C.byref(id_), C.byref(d)))
            out.append((id_.value, d.value))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib.dawn_best_free(self._h)
            self._h = None
