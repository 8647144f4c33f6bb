"""ctypes loader for the CPU oracle (oracle/libdawn_oracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from dawnsearch_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdawn_oracle.so")
EM_LEN = 384

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "dawn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


class _NodeRef(C.Structure):
    _fields_ = [("id", C.c_size_t), ("distance", C.c_float)]


class _Best(C.Structure):
    _fields_ = [("results", C.POINTER(_NodeRef)), ("len", C.c_size_t), ("worst_result_index", C.c_size_t),
                ("worst_distance", C.c_float), ("size", C.c_size_t)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    for name in ("orc_distance_l2sq", "orc_distance_ip", "orc_distance_cosine"):
        getattr(L, name).argtypes = [_f32p, _f32p]
        getattr(L, name).restype = C.c_float
    L.orc_vector_length.argtypes = [_f32p]
    L.orc_vector_length.restype = C.c_float
    L.orc_is_normalized.argtypes = [_f32p]
    L.orc_is_normalized.restype = C.c_int
    L.orc_normalize.argtypes = [_f32p, C.c_size_t]
    L.orc_to24.argtypes = [_f32p, _u8p]
    L.orc_from24.argtypes = [_u8p, _f32p]
    L.orc_from24.restype = C.c_int
    L.orc_f32_to_i16.argtypes = [C.c_float]
    L.orc_f32_to_i16.restype = C.c_int16
    L.orc_best_new.argtypes = [C.c_size_t]
    L.orc_best_new.restype = C.POINTER(_Best)
    L.orc_best_free.argtypes = [C.POINTER(_Best)]
    L.orc_best_insert.argtypes = [C.POINTER(_Best), C.c_size_t, C.c_float]
    L.orc_best_insert.restype = C.c_int
    L.orc_best_sort.argtypes = [C.POINTER(_Best)]
    L.orc_best_worst_distance.argtypes = [C.POINTER(_Best)]
    L.orc_best_worst_distance.restype = C.c_float
    L.orc_scan_topk.argtypes = [_f32p, C.c_void_p, C.c_size_t, _f32p, C.c_size_t, _u64p, _f32p]
    L.orc_scan_topk.restype = C.c_size_t
    L.orc_scan_topk_mt.argtypes = [_f32p, C.c_void_p, C.c_size_t, _f32p, C.c_size_t, _u64p, _f32p, C.c_int]
    L.orc_scan_topk_mt.restype = C.c_size_t
    L.orc_scan_topk_synth.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint64, C.c_int, _f32p, C.c_size_t, C.c_size_t,
                                      _u64p, _f32p, C.c_int]
    L.orc_scan_topk_synth.restype = C.c_size_t
    L.orc_scan_examples_old.argtypes = [_u8p, C.c_size_t, _f32p, np.ctypeslib.ndpointer(dtype=np.uintp), _f32p]
    L.orc_scan_examples_old.restype = C.c_size_t
    L.orc_scan_topk_synth_dist.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_size_t, C.c_uint64, C.c_int, _f32p, C.c_size_t,
                                           C.c_size_t] + L.orc_scan_topk_synth.argtypes[8:]
    L.orc_scan_topk_synth_dist.restype = C.c_size_t
    L.orc_synth_topical_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, _f32p]
    L.orc_splitmix64.argtypes = [C.c_uint64]
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_synth_uniform.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_synth_uniform.restype = C.c_float
    L.orc_synth_unit_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, _f32p]
    L.orc_synth_scaled.argtypes = [C.c_uint64, C.c_size_t, C.c_float, C.c_float, _f32p]
    L.orc_bert_synth.argtypes = [C.c_uint64]
    L.orc_bert_synth.restype = C.c_void_p
    L.orc_bert_synth_style.argtypes = [C.c_uint64, C.c_int]
    L.orc_bert_synth_style.restype = C.c_void_p
    L.orc_synth_scaled_normal.argtypes = [C.c_uint64, C.c_size_t, C.c_float, C.c_float, _f32p]
    L.orc_bert_free_synth.argtypes = [C.c_void_p]
    L.orc_bert_forward.argtypes = [C.c_void_p, _u32p, C.c_int, _f32p]
    L.orc_embed.argtypes = [C.c_void_p, _u32p, C.c_int, _f32p]
    L.orc_embed_padded_batch.argtypes = [C.c_void_p, _u32p, _i32p, C.c_int, C.c_uint32, _f32p]
    _lib = L
    return L


# ---- convenience wrappers -------------------------------------------------------------------

def unit_rows(seed: int, first_row: int, n: int) -> np.ndarray:
    out = np.empty((n, EM_LEN), dtype=np.float32)
    lib().orc_synth_unit_rows(seed, first_row, n, out)
    return out


def unit_rows_topical(seed: int, first_row: int, n: int, runs: bool = False) -> np.ndarray:
    """Rows of the topical mixture (synth_dist 4; runs: 5) — dawnsearch_amd/synth.py: unit_rows_topical."""
    out = np.empty((n, EM_LEN), dtype=np.float32)
    lib().orc_synth_topical_rows(seed, first_row, n, int(runs), out)
    return out


def scan_topk(x: np.ndarray, ids, q: np.ndarray, k: int, threads: int = 1):
    """Exact (distance asc, position asc) top-k. Returns (labels u64[found], distances f32[found])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n = x.shape[0]
    labels = np.zeros(max(k, 1), dtype=np.uint64)
    dist = np.zeros(max(k, 1), dtype=np.float32)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        idp = ids.ctypes.data_as(C.c_void_p)
    if threads == 1:
        found = lib().orc_scan_topk(x, idp, n, q, k, labels, dist)
    else:
        found = lib().orc_scan_topk_mt(x, idp, n, q, k, labels, dist, threads)
    return labels[:found].copy(), dist[:found].copy()


def scan_topk_synth(seed: int, first_row: int, n: int, first_id: int, Q: np.ndarray, k: int, bf16: bool = False,
                    threads: int = 0, dist: int = 0):
    """Exact top-k of every query of Q over the synthetic rows [first_row, first_row + n) of stream `seed`, generated chunk
    by chunk on all host cores (the rows are never materialised): (labels u64 [nq][found], distances f32 [nq][found])."""
    Q = np.ascontiguousarray(np.atleast_2d(Q), dtype=np.float32)
    nq = Q.shape[0]
    labels = np.zeros((nq, k), dtype=np.uint64)
    dists = np.zeros((nq, k), dtype=np.float32)
    found = lib().orc_scan_topk_synth_dist(seed, dist, first_row, n, first_id, int(bf16), Q, nq, k, labels, dists, threads)
    return labels[:, :found].copy(), dists[:, :found].copy()


class BestResults:
    """best_results.rs:28-107 through the C restatement."""

    def __init__(self, size: int):
        self._b = lib().orc_best_new(size)

    def insert(self, id_: int, distance: float) -> bool:
        return bool(lib().orc_best_insert(self._b, id_, np.float32(distance)))

    def sort(self):
        lib().orc_best_sort(self._b)

    def worst_distance(self) -> float:
        return float(lib().orc_best_worst_distance(self._b))

    def results(self):
        b = self._b.contents
        return [(int(b.results[i].id), float(b.results[i].distance)) for i in range(b.len)]

    def __del__(self):
        try:
            lib().orc_best_free(self._b)
        except Exception:
            pass


class SynthBert:
    def __init__(self, seed: int, style: int = 0):
        self._w = lib().orc_bert_synth_style(seed, style)

    def forward(self, ids: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), EM_LEN), dtype=np.float32)
        lib().orc_bert_forward(self._w, ids, len(ids), out)
        return out

    def embed(self, ids: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty(EM_LEN, dtype=np.float32)
        lib().orc_embed(self._w, ids, len(ids), out)
        return out

    def embed_padded_batch(self, seqs, pad_id: int = 0) -> np.ndarray:
        lens = np.array([len(s) for s in seqs], dtype=np.int32)
        flat = np.concatenate(seqs).astype(np.uint32)
        out = np.empty((len(seqs), EM_LEN), dtype=np.float32)
        lib().orc_embed_padded_batch(self._w, flat, lens, len(seqs), pad_id, out)
        return out

    def __del__(self):
        try:
            lib().orc_bert_free_synth(self._w)
        except Exception:
            pass
