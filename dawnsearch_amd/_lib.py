"""ctypes binding of libdawn_hip.so — the C ABI declared in include/dawn_hip.h.

The library is the product; this module only loads it.  There is no Python/torch/CPU fallback: if the
shared object is missing, import fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C dawnsearch_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DAWN_LIB: another build of the same library (tools only: the `make EXPERIMENTS=1` build with the timing-experiment kernels)
LIB_PATH = os.environ.get("DAWN_LIB") or os.path.join(_HERE, "libdawn_hip.so")

DAWN_OK = 0
ERR_INVALID_ARG, ERR_NOT_NORMALIZED, ERR_HIP, ERR_IO, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_OOM = -1, -2, -3, -4, -5, -6, -7
EM_LEN = 384
MAX_K = 64
DTYPE_F32 = 0
DTYPE_BF16 = 1


class DawnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[dawn {code}] {msg}")
        self.code = code


class NotNormalizedError(DawnError):
    """`bail!("Search vector is not normalized")` — search_provider.rs:206-208,265-267."""


def _load() -> C.CDLL:
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as
    # /opt/rocm's); whichever is mapped first serves both, and torch reports "No HIP GPUs are available" when the
    # system copy got in first.  torch is this package's plumbing for streams / torch.distributed, so when it is
    # installed let it map its runtime before libdawn_hip.so resolves the SONAME.
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent: libdawn_hip.so uses the system ROCm runtime
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. dawnsearch_amd has no fallback path; "
            "run `make -C dawnsearch_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _sz, _u64, _i32, _i64 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_int64
_pp = C.POINTER(C.c_void_p)

_SIGS = {
    "dawn_last_error": (C.c_char_p, []),
    "dawn_version": (_i32, []),
    "dawn_device_count": (_i32, [C.POINTER(_i32)]),
    "dawn_index_create": (_i32, [_sz, _i32, _i32, _pp]),
    "dawn_index_create_sharded": (_i32, [_sz, _i32, _i32, C.POINTER(_i32), _pp]),
    "dawn_index_shard_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), _vp, _sz]),
    "dawn_index_destroy": (None, [_vp]),
    "dawn_index_reserve": (_i32, [_vp, _sz]),
    "dawn_index_size": (_sz, [_vp]),
    "dawn_index_capacity": (_sz, [_vp]),
    "dawn_index_add": (_i32, [_vp, _u64, _vp]),
    "dawn_index_add_batch": (_i32, [_vp, _sz, _vp, _vp]),
    "dawn_index_search": (_i32, [_vp, _vp, _sz, _vp, _vp, C.POINTER(_sz)]),
    "dawn_index_search_batch": (_i32, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "dawn_index_search_limited": (_i32, [_vp, _vp, _sz, C.c_float, _vp, _vp, _vp]),
    "dawn_index_save": (_i32, [_vp, C.c_char_p]),
    "dawn_index_load": (_i32, [_vp, C.c_char_p]),
    "dawn_index_load_page_entries": (_i32, [_vp, C.c_char_p, _u64]),
    "dawn_index_search_device": (_i32, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "dawn_topk_merge_device": (_i32, [_i32, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dawn_result_blob_bytes": (_sz, [_sz, _sz]),
    "dawn_topk_merge_packed_device": (_i32, [_i32, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "dawn_topk_merge_host": (_i32, [_sz, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dawn_index_fill_synthetic": (_i32, [_vp, _u64, _u64, _sz, _u64]),
    "dawn_index_get_rows": (_i32, [_vp, _sz, _sz, _vp, _vp]),
    "dawn_index_profile_enable": (_i32, [_vp, _i32]),
    "dawn_index_profile_read": (_i32, [_vp, C.POINTER(_u64), C.POINTER(C.c_double)]),
    "dawn_index_stats": (_i32, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "dawn_index_stats_ext": (_i32, [_vp, _vp, _vp, _vp]),
    "dawn_index_stats_deep": (_i32, [_vp, _vp]),
    "dawn_index_stats_ladder": (_i32, [_vp, _vp, _vp, _vp]),
    "dawn_index_memory": (_i32, [_vp, _vp, _vp, _vp]),
    "dawn_index_set_option": (_i32, [_vp, C.c_char_p, _i64]),
    "dawn_index_debug_read_diag": (_i32, [_vp, _vp, _sz]),
    "dawn_index_debug_time_full_pass": (_i32, [_vp, _sz, _i32, _vp]),
    "dawn_index_debug_stream_lists": (_i32, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "dawn_index_debug_stream_bound": (_i32, [_vp, _vp]),
    "dawn_index_debug_filter_scores": (_i32, [_vp, _vp, _sz, _vp, C.POINTER(_sz)]),
    "dawn_index_debug_f6_scores": (_i32, [_vp, _vp, _sz, _vp, C.POINTER(_sz)]),
    "dawn_index_stats_batch_feedback": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "dawn_index_debug_raw_stats": (_i32, [_vp, _vp]),
    "dawn_index_debug_i6_refine": (_i32, [_vp, _sz, _vp, _vp]),
    "dawn_vec_is_normalized": (_i32, [_vp]),
    "dawn_vec_first_not_normalized": (_sz, [_vp, _sz]),
    "dawn_vec_normalize": (None, [_vp, _sz]),
    "dawn_vec_to24": (None, [_vp, _vp]),
    "dawn_vec_from24": (_i32, [_vp, _vp]),
    "dawn_best_new": (_i32, [_sz, _pp]),
    "dawn_best_free": (None, [_vp]),
    "dawn_best_insert": (_i32, [_vp, _sz, C.c_float]),
    "dawn_best_sort": (None, [_vp]),
    "dawn_best_worst_distance": (C.c_float, [_vp]),
    "dawn_best_len": (_sz, [_vp]),
    "dawn_best_get": (_i32, [_vp, _sz, C.POINTER(_sz), C.POINTER(C.c_float)]),
    "dawn_embedder_create": (_i32, [C.c_char_p, C.c_char_p, _i32, _pp]),
    "dawn_embedder_destroy": (None, [_vp]),
    "dawn_embedder_check_files": (_i32, [C.c_char_p, C.c_char_p]),
    "dawn_embedder_forward": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "dawn_embedder_forward_device": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "dawn_embedder_set_option": (_i32, [_vp, C.c_char_p, _i64]),
    "dawn_embedder_hidden_states": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "dawn_embedder_debug_op": (_i32, [_vp, _i32, _vp, _i32, _vp]),
    "dawn_embedder_debug_gemm_time": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "dawn_tokenizer_create": (_i32, [C.c_char_p, _pp]),
    "dawn_tokenizer_destroy": (None, [_vp]),
    "dawn_tokenizer_set_max_length": (_i32, [_vp, _sz]),
    "dawn_tokenizer_vocab_size": (_sz, [_vp]),
    "dawn_tokenizer_encode": (_i32, [_vp, C.c_char_p, _vp, _sz, C.POINTER(_sz)]),
    "dawn_tokenizer_encode_batch": (_i32, [_vp, C.POINTER(C.c_char_p), _sz, _vp, _sz, _vp]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here == the .so does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return (lib.dawn_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc == DAWN_OK:
        return
    msg = last_error()
    if rc == ERR_NOT_NORMALIZED:
        raise NotNormalizedError(rc, msg)
    raise DawnError(rc, msg)


def device_count() -> int:
    n = _i32(0)
    rc = lib.dawn_device_count(C.byref(n))
    return n.value if rc == DAWN_OK else 0
