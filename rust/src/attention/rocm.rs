//! Safe wrappers over `ffi` with the names and argument order of the reference's attention API
//! (`src/attention/mod.rs:16-17`): `FlashAttentionConfig`, `flash_attention`, `prefill_attention`,
//! `paged_attention`, `write_kv_to_cache`, plus the batched forms a serving loop uses.
//! NOT COMPILED in the build image (no Rust toolchain there).
//!
//! Tensors are device buffers (`DeviceTensor`): candle 0.9 has no HIP device, so there is no
//! `candle_core::Tensor` to borrow.  Deviation from the reference, on purpose: `write_kv_to_cache`
//! updates the caches in place (the reference returns rebuilt tensors).
use std::ffi::{c_int, c_void, CStr};

use super::ffi;

/// Mirrors `crate::error::Error` (src/error.rs:10-60) for the variants this path can produce.
#[derive(thiserror::Error, Debug)]
pub enum Error {
    #[error("out of KV cache blocks")]
    OutOfBlocks,
    #[error("{0}")]
    BlockIndexOutOfBounds(String),
    #[error("tensor error: {0}")]
    Tensor(String),
    #[error("configuration error: {0}")]
    Config(String),
}
pub type Result<T> = std::result::Result<T, Error>;

fn last_error() -> String {
    // SAFETY: nvllm_get_last_error never returns NULL and the string is thread-local
    unsafe { CStr::from_ptr(ffi::nvllm_get_last_error()) }.to_string_lossy().into_owned()
}

/// nvllm_status -> Error
fn check(rc: c_int) -> Result<()> {
    match rc {
        ffi::NVLLM_OK => Ok(()),
        ffi::NVLLM_ERR_CONFIG => Err(Error::Config(last_error())),
        ffi::NVLLM_ERR_OUT_OF_BLOCKS => Err(Error::OutOfBlocks),
        ffi::NVLLM_ERR_BLOCK_INDEX_OOB => Err(Error::BlockIndexOutOfBounds(last_error())),
        _ => Err(Error::Tensor(last_error())), // Unsupported / Null / Hip / Tensor
    }
}

#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum DType {
    BF16 = 0,
    F32 = 1,
}

/// A caller-owned, contiguous device buffer with the reference's row-major layout.
#[derive(Debug, Clone)]
pub struct DeviceTensor {
    pub ptr: *mut c_void,
    pub shape: Vec<usize>,
    pub dtype: DType,
}

impl DeviceTensor {
    fn dims4(&self) -> Result<(usize, usize, usize, usize)> {
        match self.shape[..] {
            [a, b, c, d] => Ok((a, b, c, d)),
            _ => Err(Error::Tensor(format!("expected 4 dims, got {:?}", self.shape))),
        }
    }
}

/// src/attention/flash.rs:26-71.  `block_size_q/kv` are kept for API parity; the GPU kernel's
/// tiling is fixed and the result does not depend on them (tests/flash_attention_test.rs:134-174).
#[derive(Debug, Clone)]
pub struct FlashAttentionConfig {
    pub block_size_q: usize,
    pub block_size_kv: usize,
    pub causal: bool,
    pub softmax_scale: f32,
}

impl FlashAttentionConfig {
    pub fn new(head_dim: usize, causal: bool) -> Self {
        Self { block_size_q: 64, block_size_kv: 64, causal, softmax_scale: 1.0 / (head_dim as f32).sqrt() }
    }
    pub fn with_block_sizes(mut self, q: usize, kv: usize) -> Self {
        self.block_size_q = q;
        self.block_size_kv = kv;
        self
    }
    pub fn with_scale(mut self, scale: f32) -> Self {
        self.softmax_scale = scale;
        self
    }
}

/// `flash_attention(query, key, value, config)` (flash.rs:85-90): q [B,Sq,Hq,D], k/v [B,Skv,Hkv,D]
/// -> `out` [B,Sq,Hq,D] (caller-allocated).  Replaces the GPU->CPU->GPU bounce of flash.rs:95-103.
pub fn flash_attention(
    query: &DeviceTensor,
    key: &DeviceTensor,
    value: &DeviceTensor,
    config: &FlashAttentionConfig,
    out: &mut DeviceTensor,
    stream: ffi::Stream,
) -> Result<()> {
    let (b, sq, hq, d) = query.dims4()?;
    let (_, skv, hkv, _) = key.dims4()?;
    let f = match query.dtype {
        DType::BF16 => ffi::nvllm_flash_attn_fwd_bf16,
        DType::F32 => ffi::nvllm_flash_attn_fwd_f32,
    };
    // SAFETY: device pointers owned by the caller, shapes as declared
    check(unsafe {
        f(query.ptr, key.ptr, value.ptr, out.ptr, std::ptr::null_mut(), b as c_int, sq as c_int,
          skv as c_int, hq as c_int, hkv as c_int, d as c_int, config.softmax_scale,
          config.causal as c_int, stream)
    })
}

/// `prefill_attention(q, k, v, num_heads, num_kv_heads, scale, causal)` (paged.rs:55-63) ->
/// `out` [B, S, Hq*D]; the mask applies only when `causal && seq_len > 1` (paged.rs:79).
#[allow(clippy::too_many_arguments)]
pub fn prefill_attention(
    query: &DeviceTensor,
    key: &DeviceTensor,
    value: &DeviceTensor,
    num_heads: usize,
    num_kv_heads: usize,
    scale: f64,
    causal: bool,
    out: &mut DeviceTensor,
    stream: ffi::Stream,
) -> Result<()> {
    let (b, s, _, d) = query.dims4()?;
    check(unsafe {
        ffi::nvllm_prefill_attention(query.ptr, key.ptr, value.ptr, out.ptr, b as c_int, s as c_int,
                                     num_heads as c_int, num_kv_heads as c_int, d as c_int, scale,
                                     causal as c_int, query.dtype as c_int, stream)
    })
}

/// Owning handle of the library's BlockTable (src/core/block.rs:183-289).
pub struct BlockTable(*mut ffi::NvllmBlockTable);

impl BlockTable {
    pub fn new(block_size: usize) -> Self {
        Self(unsafe { ffi::nvllm_block_table_new(block_size as i64) })
    }
    pub fn append_block(&mut self, block_id: usize) {
        unsafe { ffi::nvllm_block_table_append_block(self.0, block_id as i64) }
    }
    pub fn num_blocks(&self) -> usize {
        unsafe { ffi::nvllm_block_table_num_blocks(self.0) as usize }
    }
    /// block.rs:269-283
    pub fn get_slot_mapping(&self, seq_len: usize) -> Vec<usize> {
        let mut v = vec![0i64; seq_len];
        let n = unsafe { ffi::nvllm_block_table_get_slot_mapping(self.0, seq_len as i64, v.as_mut_ptr(), seq_len as i64) };
        v.truncate(n as usize);
        v.into_iter().map(|s| s as usize).collect()
    }
}

impl Drop for BlockTable {
    fn drop(&mut self) {
        unsafe { ffi::nvllm_block_table_free(self.0) }
    }
}

/// One layer's cache, `[num_blocks, block_size, Hkv, D]` for K and for V (kv_cache.rs:75-95).
pub struct LayerKVCache {
    pub key_cache: DeviceTensor,
    pub value_cache: DeviceTensor,
}

/// `paged_attention(query, kv_cache, block_table, context_len, num_heads, num_kv_heads, head_dim,
/// scale)` (paged.rs:117-126): ONE sequence -> `out` [1, T, Hq*D].  `scratch` is device memory of
/// `nvllm_paged_attention_seq_scratch_bytes` bytes.  Validation as the reference: empty table ->
/// `Error::Config` (paged.rs:133-137), bad context_len / ids -> `Error::Tensor`.
#[allow(clippy::too_many_arguments)]
pub fn paged_attention(
    query: &DeviceTensor,
    kv_cache: &LayerKVCache,
    block_table: &BlockTable,
    context_len: usize,
    num_heads: usize,
    num_kv_heads: usize,
    head_dim: usize,
    scale: f64,
    out: &mut DeviceTensor,
    scratch: *mut c_void,
    scratch_bytes: usize,
    stream: ffi::Stream,
) -> Result<()> {
    let (b, t, _, _) = query.dims4()?;
    let (num_cache_blocks, _, _, _) = kv_cache.key_cache.dims4()?;
    check(unsafe {
        ffi::nvllm_paged_attention_seq(query.ptr, b as c_int, t as c_int, kv_cache.key_cache.ptr,
                                       kv_cache.value_cache.ptr, num_cache_blocks as i64, block_table.0,
                                       context_len as i64, num_heads as c_int, num_kv_heads as c_int,
                                       head_dim as c_int, scale, query.dtype as c_int, out.ptr, scratch,
                                       scratch_bytes, stream)
    })
}

/// `write_kv_to_cache(key, value, key_cache, value_cache, slot_mapping, block_size)`
/// (paged.rs:208-215), IN PLACE.  `scratch`: device memory of `slot_mapping.len() * 8` bytes.
#[allow(clippy::too_many_arguments)]
pub fn write_kv_to_cache(
    key: &DeviceTensor,
    value: &DeviceTensor,
    key_cache: &mut DeviceTensor,
    value_cache: &mut DeviceTensor,
    slot_mapping: &[usize],
    block_size: usize,
    scratch: *mut c_void,
    scratch_bytes: usize,
    stream: ffi::Stream,
) -> Result<()> {
    let (_, s, hkv, d) = key.dims4()?;
    let (num_cache_blocks, _, _, _) = key_cache.dims4()?;
    let slots: Vec<i64> = slot_mapping.iter().map(|&x| x as i64).collect();
    check(unsafe {
        ffi::nvllm_write_kv_to_cache(key.ptr, value.ptr, s as i64, key_cache.ptr, value_cache.ptr,
                                     num_cache_blocks as i64, slots.as_ptr(), slots.len() as i64,
                                     block_size as c_int, hkv as c_int, d as c_int, key.dtype as c_int,
                                     scratch, scratch_bytes, stream)
    })
}

/// Batched paged attention for a scheduler step (no counterpart in the reference, which takes one
/// sequence per call): q [N,T,Hq,D], device `block_tables` i32 [N,max_blocks], `context_lens` i32 [N]
/// -> `out` [N,T,Hq*D].  `workspace` of `workspace_bytes(..)` bytes; never synchronises.
#[allow(clippy::too_many_arguments)]
pub fn paged_attention_batched(
    query: &DeviceTensor,
    kv_cache: &LayerKVCache,
    block_tables: *const i32,
    context_lens: *const i32,
    max_blocks_per_seq: usize,
    num_kv_heads: usize,
    block_size: usize,
    scale: f32,
    out: &mut DeviceTensor,
    workspace: *mut c_void,
    workspace_bytes: usize,
    stream: ffi::Stream,
) -> Result<()> {
    let (n, t, hq, d) = query.dims4()?;
    let (num_cache_blocks, _, _, _) = kv_cache.key_cache.dims4()?;
    check(unsafe {
        ffi::nvllm_paged_attn_bf16(query.ptr, kv_cache.key_cache.ptr, kv_cache.value_cache.ptr,
                                   block_tables, context_lens, out.ptr, workspace, workspace_bytes,
                                   n as c_int, t as c_int, hq as c_int, num_kv_heads as c_int, d as c_int,
                                   block_size as c_int, max_blocks_per_seq as c_int,
                                   num_cache_blocks as i64, scale, stream)
    })
}

pub fn workspace_bytes(n: usize, t: usize, hq: usize, hkv: usize, d: usize, bs: usize, max_blocks: usize) -> usize {
    unsafe {
        ffi::nvllm_paged_attn_workspace_bytes(n as c_int, t as c_int, hq as c_int, hkv as c_int,
                                              d as c_int, bs as c_int, max_blocks as c_int)
    }
}

/// Which kernel a batched launch of this shape runs (pure query, nvllm_paged_attn_path).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum PagedPath {
    Decode,
    K1Paged,
    Generic,
    Mid,
}

#[allow(clippy::too_many_arguments)]
pub fn paged_attention_path(n: usize, t: usize, hq: usize, hkv: usize, d: usize, bs: usize, max_blocks: usize,
                            num_cache_blocks: usize) -> Result<PagedPath> {
    let code = unsafe {
        ffi::nvllm_paged_attn_path(n as c_int, t as c_int, hq as c_int, hkv as c_int, d as c_int, bs as c_int,
                                   max_blocks as c_int, num_cache_blocks as i64, 0)
    };
    match code {
        0 => Ok(PagedPath::Decode),
        1 => Ok(PagedPath::K1Paged),
        2 => Ok(PagedPath::Generic),
        3 => Ok(PagedPath::Mid),
        e => check(e).map(|_| PagedPath::Generic),
    }
}

/// Debug pass over a launch's tables (nvllm_debug_validate_tables): fills `counts` (device, 2 x i32)
/// with the ids outside the pool among the used blocks and the context lengths outside their rows.
#[allow(clippy::too_many_arguments)]
pub fn validate_tables(block_tables: *const i32, context_lens: *const i32, num_seqs: usize,
                       max_blocks_per_seq: usize, block_size: usize, num_cache_blocks: usize,
                       counts: *mut i32, stream: ffi::Stream) -> Result<()> {
    check(unsafe {
        ffi::nvllm_debug_validate_tables(block_tables, context_lens, num_seqs as c_int, max_blocks_per_seq as c_int,
                                         block_size as c_int, num_cache_blocks as i64, counts, stream)
    })
}

/// Batched KV write: key/value [num_tokens,Hkv,D], device `slot_mapping` i64 [num_tokens].
pub fn reshape_and_cache(
    key: &DeviceTensor,
    value: &DeviceTensor,
    kv_cache: &mut LayerKVCache,
    slot_mapping: *const i64,
    block_size: usize,
    stream: ffi::Stream,
) -> Result<()> {
    let (n, hkv, d) = match key.shape[..] {
        [n, h, d] => (n, h, d),
        _ => return Err(Error::Tensor("key must be [num_tokens, Hkv, D]".into())),
    };
    let (num_cache_blocks, _, _, _) = kv_cache.key_cache.dims4()?;
    check(unsafe {
        ffi::nvllm_reshape_and_cache_bf16(key.ptr, value.ptr, kv_cache.key_cache.ptr, kv_cache.value_cache.ptr,
                                          slot_mapping, n as i64, hkv as c_int, d as c_int,
                                          block_size as c_int, num_cache_blocks as i64, stream)
    })
}
