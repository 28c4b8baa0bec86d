"""ctypes binding of include/nvllm_attn.h (libnvllm_attn.so).

Fails loudly if the library is missing: there is no fallback implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnvllm_attn.so")

NVLLM_OK = 0
NVLLM_ERR_CONFIG = -1
NVLLM_ERR_UNSUPPORTED = -2
NVLLM_ERR_NULL = -3
NVLLM_ERR_HIP = -4
NVLLM_ERR_OUT_OF_BLOCKS = -5
NVLLM_ERR_BLOCK_INDEX_OOB = -6
NVLLM_ERR_TENSOR = -7

_c = ctypes
_vp, _i, _i64, _u64, _f, _d, _sz = (_c.c_void_p, _c.c_int, _c.c_int64, _c.c_uint64, _c.c_float,
                                    _c.c_double, _c.c_size_t)
_pi64 = _c.POINTER(_c.c_int64)
_pu64 = _c.POINTER(_c.c_uint64)
_pi = _c.POINTER(_c.c_int)

# name -> (restype, argtypes); every symbol include/nvllm_attn.h declares
SIGNATURES = {
    "nvllm_flash_attn_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "nvllm_flash_attn_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "nvllm_paged_attn_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i64, _f, _vp]),
    "nvllm_paged_attn_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i64, _f, _vp]),
    "nvllm_paged_attn_bf16_noalloc": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _f, _vp]),
    "nvllm_paged_attn_noalloc_max_scratch_bytes": (_sz, []),
    "nvllm_paged_attn_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "nvllm_paged_attn_path": (_i, [_i, _i, _i, _i, _i, _i, _i, _i64, _i]),
    "nvllm_reshape_and_cache_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i64, _vp]),
    "nvllm_reshape_and_cache_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i64, _vp]),
    "nvllm_qk_norm_rope_cache_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i64, _vp]),
    "nvllm_qk_norm_rope_cache_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i64, _vp]),
    "nvllm_rope_cache_f32": (_i, [_i64, _i64, _d, _vp, _vp]),
    "nvllm_get_last_error": (_c.c_char_p, []),
    "nvllm_synchronize": (_i, []),
    "nvllm_gather_cache_blocks": (_i, [_vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp]),
    "nvllm_kv_cache_layer_bytes": (_i64, [_i64, _i64, _i64, _i64, _i]),
    "nvllm_kv_cache_total_bytes": (_i64, [_i64, _i64, _i64, _i64, _i64, _i]),
    "nvllm_debug_set_policy": (_i, [_c.c_char_p, _i]),
    "nvllm_debug_validate_tables": (_i, [_vp, _vp, _i, _i, _i, _i64, _vp, _vp]),
    "nvllm_debug_reset_policy": (None, []),
    "nvllm_device_count": (_i, []),
    "nvllm_set_device": (_i, [_i]),
    "nvllm_compute_num_blocks": (_i64, [_i64, _i64]),
    "nvllm_hash_token_block": (_u64, [_c.POINTER(_c.c_uint32), _i64, _i, _u64]),
    "nvllm_slot_mapping_last_writer": (_i64, [_pi64, _i64, _pi64]),
    "nvllm_block_table_new": (_vp, [_i64]),
    "nvllm_block_table_free": (None, [_vp]),
    "nvllm_block_table_block_size": (_i64, [_vp]),
    "nvllm_block_table_append_block": (None, [_vp, _i64]),
    "nvllm_block_table_num_blocks": (_i64, [_vp]),
    "nvllm_block_table_is_empty": (_i, [_vp]),
    "nvllm_block_table_clear": (None, [_vp]),
    "nvllm_block_table_get_block_id": (_i, [_vp, _i64, _pi64]),
    "nvllm_block_table_get_physical_block_ids": (_i64, [_vp, _pi64, _i64]),
    "nvllm_block_table_get_slot_mapping": (_i64, [_vp, _i64, _pi64, _i64]),
    "nvllm_scheduler_config_default": (None, [_vp]),
    "nvllm_scheduler_new": (_vp, [_vp, _i64, _i64]),
    "nvllm_scheduler_free": (None, [_vp]),
    "nvllm_scheduler_add_sequence": (_i, [_vp, _u64, _vp, _i64, _c.c_int32]),
    "nvllm_scheduler_remove_sequence": (_i, [_vp, _u64]),
    "nvllm_scheduler_schedule": (_i, [_vp]),
    "nvllm_scheduler_out_num_prefill": (_i64, [_vp]),
    "nvllm_scheduler_out_num_decode": (_i64, [_vp]),
    "nvllm_scheduler_out_num_preempted": (_i64, [_vp]),
    "nvllm_scheduler_out_prefill_ids": (_vp, [_vp]),
    "nvllm_scheduler_out_decode_ids": (_vp, [_vp]),
    "nvllm_scheduler_out_preempted_ids": (_vp, [_vp]),
    "nvllm_scheduler_out_prefill_chunk_sizes": (_vp, [_vp]),
    "nvllm_scheduler_out_num_prefill_tokens": (_i64, [_vp]),
    "nvllm_scheduler_out_num_decode_tokens": (_i64, [_vp]),
    "nvllm_scheduler_finish_sequence": (None, [_vp, _u64, _i]),
    "nvllm_scheduler_append_token": (_i, [_vp, _u64, _c.c_uint32]),
    "nvllm_scheduler_mark_prefilled": (_i, [_vp, _u64, _i64]),
    "nvllm_scheduler_num_waiting": (_i64, [_vp]),
    "nvllm_scheduler_num_running": (_i64, [_vp]),
    "nvllm_scheduler_num_swapped": (_i64, [_vp]),
    "nvllm_scheduler_has_unfinished_sequences": (_i, [_vp]),
    "nvllm_scheduler_has_pending_requests": (_i, [_vp]),
    "nvllm_scheduler_running_sequence_ids": (_i64, [_vp, _vp, _i64]),
    "nvllm_scheduler_reset": (None, [_vp]),
    "nvllm_scheduler_seq_status": (_i, [_vp, _u64]),
    "nvllm_scheduler_seq_finish_reason": (_i, [_vp, _u64]),
    "nvllm_scheduler_seq_set_status": (_i, [_vp, _u64, _i]),
    "nvllm_scheduler_seq_info": (_i64, [_vp, _u64, _i]),
    "nvllm_scheduler_seq_block_ids": (_i64, [_vp, _u64, _vp, _i64]),
    "nvllm_scheduler_seq_output_tokens": (_i64, [_vp, _u64, _vp, _i64]),
    "nvllm_scheduler_num_free_blocks": (_i64, [_vp]),
    "nvllm_scheduler_num_used_blocks": (_i64, [_vp]),
    "nvllm_scheduler_num_cached_prefixes": (_i64, [_vp]),
    "nvllm_scheduler_block_ref_count": (_i64, [_vp, _i64]),
    "nvllm_scheduler_fill_decode_batch": (_i, [_vp, _vp]),
    "nvllm_scheduler_fill_prefill_batch": (_i64, [_vp, _i64, _vp, _vp, _i64]),
    "nvllm_batch_new": (_vp, [_i64]),
    "nvllm_batch_free": (None, [_vp]),
    "nvllm_batch_clear": (None, [_vp]),
    "nvllm_batch_add_sequence": (_i, [_vp, _vp, _i64, _i64]),
    "nvllm_batch_num_seqs": (_i64, [_vp]),
    "nvllm_batch_num_tokens": (_i64, [_vp]),
    "nvllm_batch_max_blocks_per_seq": (_i64, [_vp]),
    "nvllm_batch_block_tables": (_vp, [_vp]),
    "nvllm_batch_context_lens": (_vp, [_vp]),
    "nvllm_batch_slot_mapping": (_vp, [_vp]),
    "nvllm_batch_query_start": (_vp, [_vp]),
    "nvllm_batch_positions": (_vp, [_vp]),
    "nvllm_batch_upload": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "nvllm_block_manager_new": (_vp, [_i64, _i64]),
    "nvllm_block_manager_free_handle": (None, [_vp]),
    "nvllm_block_manager_block_size": (_i64, [_vp]),
    "nvllm_block_manager_num_blocks": (_i64, [_vp]),
    "nvllm_block_manager_num_free_blocks": (_i64, [_vp]),
    "nvllm_block_manager_num_used_blocks": (_i64, [_vp]),
    "nvllm_block_manager_has_free_blocks": (_i, [_vp]),
    "nvllm_block_manager_can_allocate": (_i, [_vp, _i64]),
    "nvllm_block_manager_allocate": (_i, [_vp, _pi64]),
    "nvllm_block_manager_allocate_many": (_i, [_vp, _i64, _pi64]),
    "nvllm_block_manager_free": (_i, [_vp, _i64]),
    "nvllm_block_manager_free_many": (_i64, [_vp, _pi64, _i64]),
    "nvllm_block_manager_ref_count": (_i64, [_vp, _i64]),
    "nvllm_block_manager_prefix_hash": (_i, [_vp, _i64, _pu64]),
    "nvllm_block_manager_increment_ref": (_i64, [_vp, _i64]),
    "nvllm_block_manager_decrement_ref": (_i64, [_vp, _i64]),
    "nvllm_block_manager_get_cached_block": (_i64, [_vp, _u64]),
    "nvllm_block_manager_cache_block": (None, [_vp, _i64, _u64]),
    "nvllm_block_manager_allocate_with_prefix": (_i, [_vp, _u64, _pi64, _pi]),
    "nvllm_block_manager_is_prefix_cached": (_i, [_vp, _u64]),
    "nvllm_block_manager_num_cached_prefixes": (_i64, [_vp]),
    "nvllm_block_manager_clear_prefix_cache": (None, [_vp]),
    "nvllm_block_manager_reset": (None, [_vp]),
    "nvllm_paged_attention_seq_scratch_bytes": (_sz, [_i64, _i, _i, _i, _i, _i]),
    "nvllm_paged_attention_seq": (_i, [_vp, _i, _i, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _d, _i, _vp, _vp, _sz, _vp]),
    "nvllm_write_kv_to_cache": (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _pi64, _i64, _i, _i, _i, _i, _vp, _sz, _vp]),
    "nvllm_prefill_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _i, _i, _vp]),
}

_lib = None


def lib():
    """Load libnvllm_attn.so and bind every declared symbol (raises if missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libnvllm_attn.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no fallback implementation)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().nvllm_get_last_error().decode(errors="replace")
