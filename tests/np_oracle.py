"""Independent numpy restatement of the scan/top-k/codec semantics — used ONLY to cross-check the C
oracle (oracle/dawn_oracle.c) in the CPU test-suite.  Written from the reference sources, not from the
C file: src/search/vector.rs, src/search/best_results.rs, examples_old/search.rs."""
from __future__ import annotations

import numpy as np

EM_LEN = 384
I24_MAX = 0x7FFFFF


def seq_dot(q: np.ndarray, X: np.ndarray) -> np.ndarray:
    """Sequential f32 sum of f32 products per row (vector.rs:99-101,128-134)."""
    acc = np.zeros(X.shape[0], dtype=np.float32)
    for i in range(X.shape[1]):
        acc = (acc + (X[:, i] * q[i]).astype(np.float32)).astype(np.float32)
    return acc


def distances(q, X):
    return (np.float32(1.0) - seq_dot(q, X)).astype(np.float32)


def scan_topk(X, ids, q, k):
    d = distances(q, X)
    order = np.lexsort((np.arange(len(d)), d))[:k]  # (distance asc, position asc)
    return ids[order], d[order]


def vector_length(v):
    s = np.float32(0)
    for x in v.astype(np.float32):
        dlt = np.float32(x - np.float32(0))
        s = np.float32(s + np.float32(dlt * dlt))
    return np.float32(np.sqrt(s))


def is_normalized(v):
    l = vector_length(v)
    if not np.isfinite(l):
        return False
    return bool(l > np.float32(1.0) - np.float32(0.01) and l < np.float32(1.0) + np.float32(0.01))


def to24(v):
    out = bytearray()
    for x in v.astype(np.float32):
        iv = int(((float(x) + 1.0) / 2.0) * float(I24_MAX))  # `as i32` truncates toward zero
        out += bytes((iv & 0xFF, (iv >> 8) & 0xFF, (iv >> 16) & 0xFF))
    return bytes(out)


def from24(data):
    out = np.zeros(EM_LEN, dtype=np.float32)
    for i in range(EM_LEN):
        v = data[i * 3] | (data[i * 3 + 1] << 8) | (data[i * 3 + 2] << 16)
        if data[i * 3 + 2] & 0x80:
            v |= 0xFF
        out[i] = np.float32(v / I24_MAX * 2.0 - 1.0)
    return out


class BestResults:
    """best_results.rs:28-107, transcribed behaviour (not code)."""

    def __init__(self, size):
        self.results = []
        self.worst_result_index = 0
        self.worst_distance = np.float32(0)
        self.size = size

    def _update_worst(self):
        self.worst_result_index = 0
        self.worst_distance = self.results[0][1]
        for i in range(1, len(self.results)):
            if self.results[i][1] > self.worst_distance:
                self.worst_distance = self.results[i][1]
                self.worst_result_index = i

    def insert(self, id_, d):
        d = np.float32(d)
        if len(self.results) < self.size:
            if any(r[0] == id_ for r in self.results):
                return False
            self.results.append((id_, d))
            if len(self.results) == self.size:
                self._update_worst()
            return True
        if d < self.worst_distance:
            if any(r[0] == id_ for r in self.results):
                return False
            self.results[self.worst_result_index] = (id_, d)
            self._update_worst()
            return True
        return False

    def sort(self):
        if not self.results:
            return
        self.results.sort(key=lambda r: r[1])  # Python's sort is stable, like Vec::sort_by
        self.worst_result_index = len(self.results) - 1
        self.worst_distance = self.results[-1][1]
