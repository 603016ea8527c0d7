"""ctypes binding of the C ABI in ``include/nvt_hip.h`` (``libnvt_hip.so``).

The north star asks for cffi; cffi is not installed in this image, so the thin
FFI layer is stdlib ``ctypes`` over the same ``extern "C"`` entry points.

There is deliberately NO fallback: if the shared library is missing or a GPU is
not visible, every compute entry raises.  (The oracle under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# NVT_HIP_LIB: an alternative build of the same library (A / B measurements of kernel variants)
LIB_PATH = os.environ.get("NVT_HIP_LIB") or os.path.join(_HERE, "libnvt_hip.so")

# dtype codes (include/nvt_hip.h)
NVT_F32, NVT_F64, NVT_I32, NVT_I64, NVT_U8 = 0, 1, 2, 3, 4
NVT_GB_SUMSQ, NVT_GB_MINMAX = 1, 2
NVT_EINVAL, NVT_EHIP, NVT_ENOMEM, NVT_EUNSUPPORTED = -1, -2, -3, -4   # include/nvt_hip.h
ENCODE_HEAD_BYTES = 12288 * 12 + 64   # NVT_ENCODE_HEAD_BYTES
ST_NULLS, ST_SENTINEL, ST_OCCUPIED, ST_OVERFLOW, ST_ROWS = 0, 1, 2, 3, 4
ST_MAXCOUNT = 8
ST_BIG = 9
ST_NEED = 10
STATE_WORDS = 16

_vp, _u64, _i64, _i32, _u32, _dbl = (
    C.c_void_p,
    C.c_uint64,
    C.c_int64,
    C.c_int,
    C.c_uint32,
    C.c_double,
)
_pp = C.POINTER(C.c_void_p)

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "nvt_version": [],
    "nvt_last_error": [],
    "nvt_count_table_bytes": [_i32, _u64, C.POINTER(_u64)],
    "nvt_count_clear": [_vp, _i32, _u64, _vp, _vp],
    "nvt_count_i32": [_vp, _vp, _u64, _vp, _u64, _vp, _vp],
    "nvt_count_i64": [_vp, _vp, _u64, _vp, _u64, _vp, _vp],
    "nvt_count_merge_i32": [_vp, _vp, _u64, _vp, _u64, _vp, _vp],
    "nvt_count_merge_i64": [_vp, _vp, _u64, _vp, _u64, _vp, _vp],
    "nvt_count_compact_i32": [_vp, _u64, _vp, _vp, _vp, _vp],
    "nvt_count_compact_i64": [_vp, _u64, _vp, _vp, _vp, _vp],
    "nvt_dense_count_ws_bytes": [_i32, _u64, _i32, _i32, C.POINTER(_u64)],
    "nvt_range_table_bytes": [_i32, C.POINTER(_u64)],
    "nvt_dense_count_i32": [_vp, _vp, _vp, _u64, _i32, _vp, _vp, _vp, _u64, _vp, _vp],
    "nvt_dense_count_i64": [_vp, _vp, _vp, _u64, _i32, _vp, _vp, _vp, _u64, _vp, _vp],
    "nvt_vocab_sort_tmp_bytes": [_i32, _u64, C.POINTER(_u64)],
    "nvt_vocab_order_tmp_bytes": [_u64, _u64, C.POINTER(_u64)],
    "nvt_class_hist": [_vp, _u64, _vp, _vp],
    "nvt_vocab_sort_i32": [_vp, _vp, _u64, _i64, _vp, _vp],
    "nvt_vocab_sort_i64": [_vp, _vp, _u64, _i64, _vp, _vp],
    "nvt_encode_table_bytes": [_i32, _u64, C.POINTER(_u64)],
    "nvt_encode_build_i32": [_vp, _u64, _i64, _vp, _u64, _vp, _i32, _vp],
    "nvt_encode_build_i64": [_vp, _u64, _i64, _vp, _u64, _vp, _i32, _vp],
    "nvt_encode_i32": [_vp, _vp, _u64, _vp, _u64, _vp, _i64, _i64, _u32, _vp, _i32, _vp, _u64, _i64,
                       _vp],
    "nvt_encode_i64": [_vp, _vp, _u64, _vp, _u64, _vp, _i64, _i64, _u32, _vp, _i32, _vp, _u64, _i64,
                       _vp],
    "nvt_hash_bucket_i32": [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp],
    "nvt_hash_bucket_i64": [_vp, _vp, _u64, _u32, _vp, _vp, _vp, _vp],
    "nvt_moments_scratch_bytes": [],
    "nvt_moments": [_vp, _i32, _vp, _u64, _i32, _dbl, _vp, _vp, _vp],
    "nvt_minmax": [_vp, _i32, _vp, _u64, _i32, _vp, _vp, _vp],
    "nvt_fill_normalize": [_vp, _i32, _vp, _u64, _i32, _dbl, _i32, _dbl, _dbl, _vp, _i32, _vp, _vp],
    "nvt_clip_log": [_vp, _i32, _vp, _u64, _i32, _dbl, _i32, _dbl, _i32, _dbl, _i32, _vp, _i32, _vp],
    "nvt_bucketize": [_vp, _i32, _vp, _u64, _vp, _i32, _vp, _vp],
    "nvt_gb_create": [_i32, _i32, _i32, _u64, _pp],
    "nvt_gb_destroy": [_vp],
    "nvt_gb_table_bytes": [_i32, _i32, _i32, _u64, C.POINTER(_u64)],
    "nvt_gb_create_in": [_i32, _i32, _i32, _u64, _vp, _u64, _pp],
    "nvt_gb_state_ptr": [_vp],
    "nvt_gb_update_ws_bytes": [_u64, C.POINTER(_u64)],
    "nvt_gb_set_workspace": [_vp, _vp, _u64],
    "nvt_gb_clear": [_vp, _vp],
    "nvt_gb_update": [_vp, _pp, _pp, _pp, C.POINTER(C.c_int), _pp, _u64, _vp],
    "nvt_gb_merge": [_vp, _pp, _vp, _vp, _vp, _pp, _pp, _pp, _pp, _u64, _vp],
    "nvt_gb_state": [_vp, C.POINTER(_u64), _vp],
    "nvt_gb_compact": [_vp, _pp, _vp, _vp, _vp, _pp, _pp, _pp, _pp, _vp, _vp],
    "nvt_gb_index_build": [_vp, _pp, _vp, _u64, _vp],
    "nvt_gb_lookup": [_vp, _pp, _pp, _u64, _vp, _vp],
    "nvt_sort_key_u64": [_vp, _i32, _vp, _u64, _i32, _vp, _vp],
    "nvt_order_rows_ws_bytes": [_u64, C.POINTER(_u64)],
    "nvt_order_rows": [_vp, _vp, _u64, _vp, _u64, _vp, _vp, _vp],
    "nvt_seg_aggregate": [_vp, _u64, _u64, _pp, C.POINTER(C.c_int), _pp, _i32, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp],
    "nvt_gather_f64": [_vp, _vp, _u64, _dbl, _vp, _i32, _vp],
    "nvt_te_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _dbl, _dbl, _vp, _i32, _vp],
    "nvt_te_apply_folds": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _u64, _dbl, _dbl, _vp, _i32, _vp],
    "nvt_sgb_sort_ws_bytes": [_u64, C.POINTER(_u64)],
    "nvt_key_minmax": [_vp, _i32, _u64, _vp, _vp],
    "nvt_sgb_sort": [_vp, _i32, _i64, _vp, _i32, _u64, _vp, C.POINTER(_vp), C.POINTER(C.c_int), _vp],
    "nvt_sgb_regroup_ws_bytes": [_u64, C.POINTER(_u64)],
    "nvt_sgb_regroup": [_vp, _i32, _i32, _i64, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp],
    "nvt_sgb_reduce": [_vp, _i32, _i32, _pp, C.POINTER(C.c_int), _pp, _i32, _i32, _u64, _u64,
                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _pp, _pp, _vp],
    "nvt_flat_lookup_gather": [_vp, _i32, _vp, _u64, _vp, _vp, _u64, _i64, _vp, _i32, _pp,
                               C.POINTER(C.c_int), C.POINTER(_dbl), _vp, _vp],
    "nvt_flat_lookup_te": [_vp, _i32, _vp, _u64, _vp, _vp, _u64, _i64, _vp, _i32, _vp, _dbl, _dbl, _vp,
                           _i32, _vp],
    "nvt_flat_lookup_image": [_vp, _i32, _vp, _u64, _vp, _vp, _u64, _i64, _vp, _vp, _vp, _u32, _i32, _pp,
                              _pp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u64), _vp, _vp],
    "nvt_image_pack": [_pp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_u32), _i32, _u64, _vp,
                       _u32, _vp],
    "nvt_jg_image": [_vp, _pp, _pp, _pp, _pp, _i32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                     C.POINTER(_u32), _i32, _u64, _vp, _u32, _vp],
    "nvt_te_image": [_vp, _vp, _vp, _vp, _i32, _u64, _dbl, _dbl, _i32, _vp, _u32, _u32, _vp],
    "nvt_fold_mt19937_par_ws_bytes": [_u64, _i32, C.POINTER(_u64)],
    "nvt_fold_mt19937_par": [_u32, _i32, _u64, _vp, _vp, _u64, _vp, _vp],
    "nvt_encode_stats": [C.POINTER(_u64), _i32, _vp],
    "nvt_pq_decode_chunk": [_vp, _u64, _i32, _i32, _u64, _vp, _u64, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64)],
    "nvt_pq_decode_chunk_codec": [_vp, _u64, _i32, _i32, _i32, _u64, _vp, _u64, _vp, _u64, _vp, _u64,
                                  C.POINTER(_u64), C.POINTER(_u64)],
    "nvt_expand_valid_ws_bytes": [_u64, C.POINTER(_u64)],
    "nvt_expand_valid": [_vp, _i32, _vp, _u64, _vp, _vp, _vp],
    "nvt_exchange_ranges": [_vp, _i32, _vp, _vp],
    "nvt_exchange_ranges_sorted": [_vp, _i32, _vp, _vp],
    "nvt_exchange_hist": [_vp, _i32, C.POINTER(_i64), C.POINTER(_u64), _i32, _vp, _vp],
    "nvt_exchange_scatter": [_vp, _i32, C.POINTER(_i64), C.POINTER(_u64), _i32, _vp, _vp, _vp],
    "nvt_exchange_pack_ordered": [_vp, _i32, C.POINTER(_i64), C.POINTER(_u64), _i32, _vp, _vp, _vp, _vp],
    "nvt_exchange_unpack": [_vp, _u64, _vp, _vp, _i32, _vp, _vp, _vp],
    "nvt_count_merge_sorted_ws_bytes": [_u64, C.POINTER(_u64)],
    "nvt_count_merge_sorted": [_vp, _u64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "nvt_flat_index_tmp_bytes": [_u64, C.POINTER(_u64)],
    "nvt_flat_index_build": [_vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp],
    "nvt_flat_lookup": [_vp, _i32, _vp, _u64, _vp, _vp, _u64, _i64, _vp, _vp],
    "nvt_widen_i64": [_vp, _i32, _u64, _vp, _vp],
    "nvt_popcount": [_vp, _u64, _vp, _vp],
    "nvt_fold_mt19937": [_u32, _i32, _u64, _vp, _vp],
    "nvt_prefix_distinct": [_vp, _i32, _vp, _vp],
    "nvt_vocab_label_shard": [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nvt_exchange_unpack2": [_vp, _vp, _u64, _vp, _vp, _i32, _vp, _vp, _vp, _vp],
}


class PrefixCol(C.Structure):
    _fields_ = [("keys", _vp), ("valid", _vp), ("n", _u64), ("key_bytes", _i32)]


class XCol(C.Structure):
    _fields_ = [("keys", _vp), ("counts", _vp), ("n", _u64)]


class MomentsCol(C.Structure):
    _fields_ = [("x", _vp), ("valid", _vp), ("n", _u64), ("dtype", C.c_int32),
                ("has_fill", C.c_int32), ("fill_val", _dbl), ("out3", _vp)]


class FillNormCol(C.Structure):
    _fields_ = [("x", _vp), ("valid", _vp), ("n", _u64), ("dtype", C.c_int32),
                ("has_fill", C.c_int32), ("fill_val", _dbl), ("do_norm", C.c_int32),
                ("out_dtype", C.c_int32), ("shift", _dbl), ("scale", _dbl), ("out", _vp),
                ("filled", _vp), ("moments", _vp)]


class CountCol(C.Structure):
    _fields_ = [("keys", _vp), ("valid", _vp), ("weights", _vp), ("n", _u64),
                ("key_bytes", C.c_int32), ("path", C.c_int32), ("ws", _vp), ("out_keys", _vp),
                ("out_counts", _vp), ("out_capacity", _u64), ("state", _vp), ("hot_image", _vp),
                ("range_table", _vp)]


class VocabCol(C.Structure):
    _fields_ = [("keys", _vp), ("counts", _vp), ("n", _u64), ("max_count", _i64),
                ("key_bytes", C.c_int32), ("unique_keys", C.c_int32), ("sort_tmp", _vp),
                ("first_label", _i64), ("table", _vp), ("capacity", _u64),
                ("sentinel_label", _vp), ("ready_event", _vp), ("src_keys", _vp),
                ("src_counts", _vp), ("cls_hist", _vp), ("n_big", _u64), ("range_aux", _vp),
                ("range_nb_log2", C.c_int32), ("flat_slots", C.c_uint64), ("src_labels", _vp),
                ("head_image", _vp)]


class MergeCol(C.Structure):
    _fields_ = [("a_keys", _vp), ("a_counts", _vp), ("na", _u64), ("b_keys", _vp),
                ("b_counts", _vp), ("nb", _u64), ("out_keys", _vp), ("out_counts", _vp),
                ("src_a", _vp), ("src_b", _vp), ("out_n", _vp)]


class ImagePart(C.Structure):
    """nvt_image_part: one operator's byte range of the records (nvt_image_build)."""
    _fields_ = [("kind", C.c_int32), ("nvals", C.c_int32), ("ncols", C.c_int32), ("kfold", C.c_int32),
                ("out_dtype", C.c_int32), ("offset", C.c_uint32), ("groups", _u64), ("count", _vp),
                ("sum", _vp), ("sumsq", _vp), ("mn", _vp), ("mx", _vp), ("kinds", _vp), ("vals", _vp),
                ("dst_dtypes", _vp), ("offs", _vp), ("tot_count", _vp), ("tot_sum", _vp),
                ("fold_count", _vp), ("fold_sum", _vp), ("p_smooth", _dbl), ("y_mean", _dbl),
                ("moments", _vp)]


IMAGE_PART_JG, IMAGE_PART_TE = 0, 1   # include/nvt_hip.h NVT_IMAGE_PART_*
IMAGE_BUILD_MAX_PARTS, IMAGE_BUILD_MAX_STRIDE = 4, 192


class EncodeCol(C.Structure):
    _fields_ = [("keys", _vp), ("valid", _vp), ("n", _u64), ("table", _vp), ("capacity", _u64),
                ("sentinel_label", _vp), ("null_label", _i64), ("oov_label", _i64),
                ("num_buckets", _u32), ("key_bytes", C.c_int32), ("out_bytes", C.c_int32),
                ("out", _vp), ("vocab_keys", _vp), ("n_vocab", _u64), ("first_label", _i64),
                ("wait_event", _vp), ("range_aux", _vp), ("head_image", _vp)]


SIGNATURES.update({
    "nvt_keydir_build": [_vp, _u64, _u64, _vp, _vp],
    "nvt_keydir_lookup_image": [_vp, _i32, _vp, _u64, _vp, _u64, _vp, _u64, _i64, _i64, _vp, _u32, _i32, _pp,
                                _pp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u64), _vp, _vp],
    "nvt_image_build": [C.POINTER(ImagePart), _i32, _u64, _vp, _u32, _vp],
    "nvt_moments_many": [C.POINTER(MomentsCol), _i32, _vp, _vp],
    "nvt_fill_normalize_many": [C.POINTER(FillNormCol), _i32, _vp],
    "nvt_dense_count_many": [C.POINTER(CountCol), _i32, _vp],
    "nvt_vocab_finalize_many": [C.POINTER(VocabCol), _i32, _vp],
    "nvt_encode_many": [C.POINTER(EncodeCol), _i32, _vp],
    "nvt_merge_sorted_ws_bytes": [C.POINTER(MergeCol), _i32, C.POINTER(_u64)],
    "nvt_merge_sorted_many": [C.POINTER(MergeCol), _i32, _vp, _u64, _vp],
    "nvt_merge_payload": [_vp, _vp, _u64, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "nvt_event_create": [_pp],
    "nvt_event_destroy": [_vp],
    "nvt_stream_wait_event": [_vp, _vp],
    "nvt_mailbox_create": [_u64, _pp],
    "nvt_mailbox_destroy": [_vp],
    "nvt_mailbox_data": [_vp],
    "nvt_mailbox_capacity": [_vp],
    "nvt_mailbox_post": [_vp, _vp, _u64, _vp, C.POINTER(_u64)],
    "nvt_mailbox_wait": [_vp, _u64, _dbl],
    "nvt_prof_begin": [],
    "nvt_prof_report": [C.c_char_p, _u64, C.POINTER(_u64)],
    "nvt_range_push": [C.c_char_p],
    "nvt_range_pop": [],
})

_RESTYPES = {
    "nvt_last_error": C.c_char_p,
    "nvt_moments_scratch_bytes": C.c_uint64,
    "nvt_gb_destroy": None,
    "nvt_range_push": None,
    "nvt_range_pop": None,
    "nvt_event_destroy": None,
    "nvt_mailbox_destroy": None,
    "nvt_mailbox_data": C.c_void_p,
    "nvt_gb_state_ptr": C.c_void_p,
    "nvt_mailbox_capacity": C.c_uint64,
}

_lock = threading.Lock()
_lib = None


class NvtHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libnvt_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NvtHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "nvtabular_amd/csrc`). There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().nvt_last_error()
        raise NvtHipError(f"{what or 'nvt call'} failed ({rc}): {msg.decode() if msg else ''}")


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise NvtHipError(
            "no MI355X visible (torch.cuda.is_available() is False); the NVTabular hot path "
            "has no CPU fallback in this engine"
        )


def ptr_array(ptrs):
    """Host array of device pointers (void*[n]); None -> NULL."""
    arr = (C.c_void_p * max(len(ptrs), 1))()
    for i, p in enumerate(ptrs):
        arr[i] = p
    return arr
