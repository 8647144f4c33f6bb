"""Host WordPiece tokenizer over the C ABI (dawn_tokenizer_*): the `tokenizers` crate calls of the reference's
EmbeddingProvider (src/embedding/embedding_service.rs:88,101-113).  Pure host code."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import check, lib


class Tokenizer:
    """`Tokenizer::from_file(tokenizer.json | vocab.txt)`; `encode(text)` -> ids incl. [CLS]/[SEP]."""

    def __init__(self, path: str, max_length: int | None = None):
        h = C.c_void_p()
        check(lib.dawn_tokenizer_create(path.encode(), C.byref(h)))
        self._h = h
        if max_length is not None:
            check(lib.dawn_tokenizer_set_max_length(self._h, max_length))

    def close(self):
        if getattr(self, "_h", None):
            lib.dawn_tokenizer_destroy(self._h)
            self._h = None

    __del__ = close

    def vocab_size(self) -> int:
        return lib.dawn_tokenizer_vocab_size(self._h)

    def encode(self, text: str) -> np.ndarray:
        raw = text.encode("utf-8", "surrogatepass").replace(b"\x00", b"")  # C strings end at NUL; clean_text drops it
        cap = 4 * len(raw) + 8
        out = np.zeros(cap, dtype=np.uint32)
        n = C.c_size_t(0)
        check(lib.dawn_tokenizer_encode(self._h, raw, C.c_void_p(out.ctypes.data), cap, C.byref(n)))
        return out[:n.value].copy()

    def encode_batch(self, texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
        """-> (packed ids u32, seq_offsets i32 [B+1]) ready for dawn_embedder_forward."""
        raws = [t.encode("utf-8", "surrogatepass").replace(b"\x00", b"") for t in texts]
        arr = (C.c_char_p * len(raws))(*raws)
        cap = sum(4 * len(r) + 8 for r in raws) + 8
        out = np.zeros(cap, dtype=np.uint32)
        offs = np.zeros(len(raws) + 1, dtype=np.int32)
        check(lib.dawn_tokenizer_encode_batch(self._h, arr, len(raws), C.c_void_p(out.ctypes.data), cap,
                                              C.c_void_p(offs.ctypes.data)))
        return out[:offs[-1]].copy(), offs

    def __call__(self, text: str) -> List[int]:
        return self.encode(text).tolist()
