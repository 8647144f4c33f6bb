"""Host-side mirror of `EmbeddingProvider` (src/embedding/embedding_service.rs:49-139) over the C ABI.

`calculate_embedding(inputs)` keeps the reference's name and meaning (batch of inputs -> one unit vector of
384 f32 each).  Token ids cross the ABI (the tokenizer is host-side: SURVEY §8f rank 1), so `inputs` are
token-id sequences unless a `tokenizer` callable (text -> ids incl. [CLS]/[SEP]) was given.
Unlike the reference's padded batch (BatchLongest + no mask, :101-128), every input gets its batch-1 result.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import tempfile
from typing import Callable, Optional, Sequence

import numpy as np

from . import synth
from ._lib import EM_LEN, check, lib


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _pack(seqs: Sequence[np.ndarray]):
    lens = [len(s) for s in seqs]
    offs = np.zeros(len(seqs) + 1, dtype=np.int32)
    offs[1:] = np.cumsum(lens)
    flat = (np.concatenate([np.asarray(s, dtype=np.uint32) for s in seqs]) if seqs
            else np.zeros(0, dtype=np.uint32))
    return np.ascontiguousarray(flat, dtype=np.uint32), offs


class EmbeddingProvider:
    def __init__(self, safetensors_path: str, config_json_path: Optional[str] = None, device: int = 0,
                 tokenizer: Optional[Callable[[str], Sequence[int]]] = None):
        h = C.c_void_p()
        check(lib.dawn_embedder_create(safetensors_path.encode(),
                                       config_json_path.encode() if config_json_path else None, device, C.byref(h)))
        self._h = h
        self.device = device
        self.tokenizer = tokenizer

    def close(self):
        if getattr(self, "_h", None):
            lib.dawn_embedder_destroy(self._h)
            self._h = None

    __del__ = close

    def _ids(self, inputs):
        if inputs and isinstance(inputs[0], str):
            if self.tokenizer is None:
                raise ValueError("text inputs need a tokenizer (token ids cross the C ABI)")
            return [np.asarray(self.tokenizer(t), dtype=np.uint32) for t in inputs]
        return [np.asarray(s, dtype=np.uint32) for s in inputs]

    def calculate_embedding(self, inputs) -> np.ndarray:
        """embedding_service.rs:97-139 -> [len(inputs), 384] f32 unit vectors."""
        seqs = self._ids(inputs)
        flat, offs = _pack(seqs)
        out = np.zeros((len(seqs), EM_LEN), dtype=np.float32)
        check(lib.dawn_embedder_forward(self._h, _ptr(flat), _ptr(offs), len(seqs), _ptr(out)))
        return out

    def hidden_states(self, inputs):
        """BertModel::forward (model.rs:565-570): list of [S_i, 384] arrays."""
        seqs = self._ids(inputs)
        flat, offs = _pack(seqs)
        out = np.zeros((int(offs[-1]), EM_LEN), dtype=np.float32)
        check(lib.dawn_embedder_hidden_states(self._h, _ptr(flat), _ptr(offs), len(seqs), _ptr(out)))
        return [out[offs[i]:offs[i + 1]] for i in range(len(seqs))]

    def set_option(self, name: str, value: int):
        check(lib.dawn_embedder_set_option(self._h, name.encode(), value))

    def debug_op(self, op: int, data: np.ndarray, T: int, out_cols: int = EM_LEN) -> np.ndarray:
        """Test hook (dawn_embedder_debug_op): one kernel of the forward in isolation."""
        data = np.ascontiguousarray(data)
        out = np.zeros((T, out_cols), dtype=np.float32)
        check(lib.dawn_embedder_debug_op(self._h, op, _ptr(data), T, _ptr(out)))
        return out

    def forward_device(self, d_token_ids: int, d_seq_offsets: int, B: int, total_tokens: int, max_len: int,
                       d_out: int, stream: int = 0):
        check(lib.dawn_embedder_forward_device(self._h, d_token_ids, d_seq_offsets, B, total_tokens, max_len, d_out,
                                               stream))


def write_synthetic_model(dirpath: str, seed: int = 3, prefix: str = "", gamma_beta: bool = False, style: int = 0):
    """Write model.safetensors + config.json with the seeded synthetic weights (style 1: the "wide"
    bell-shaped weights with large LayerNorm gains of synth.bert_tensor_specs_wide)."""
    from safetensors.numpy import save_file
    w = synth.bert_weights(seed, style=style)
    out = {}
    for k, v in w.items():
        if gamma_beta and ".LayerNorm." in k:
            k = k.replace(".LayerNorm.weight", ".LayerNorm.gamma").replace(".LayerNorm.bias", ".LayerNorm.beta")
        out[prefix + k] = np.ascontiguousarray(v)
    os.makedirs(dirpath, exist_ok=True)
    st = os.path.join(dirpath, "model.safetensors")
    cj = os.path.join(dirpath, "config.json")
    save_file(out, st, metadata={"format": "pt"})
    with open(cj, "w") as f:
        json.dump(synth.MINILM_CONFIG, f)
    return st, cj


def smoke_check():
    """Used by __graft_entry__.smoke(): tiny forward on cuda:0 vs the committed golden fixture."""
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    g = np.load(os.path.join(gold, "minilm_seed3.npz"))
    with tempfile.TemporaryDirectory() as d:
        st, cj = write_synthetic_model(d, seed=3)
        ep = EmbeddingProvider(st, cj, 0)
        offs, toks = g["seq_offsets"], g["token_ids"]
        seqs = [toks[offs[b]:offs[b + 1]] for b in range(4)]
        emb = ep.calculate_embedding(seqs)
        err = float(np.abs(emb - g["embeddings"][:4]).max())
        assert err < 1e-5, err
        ep.close()
    return err
