//! Patch of the reference's `src/attention/mod.rs` (:13-17): the CPU modules stay, the MI355X
//! path is added behind the `rocm` feature.  NOT COMPILED in the build image.
//!
//! ```ignore
//! pub mod flash;   // reference, unchanged (the CPU oracle of this path)
//! pub mod paged;   // reference, unchanged
//! pub use flash::{FlashAttentionConfig, flash_attention, flash_attention_cpu};
//! pub use paged::{paged_attention, prefill_attention, write_kv_to_cache};
//! ```
#[cfg(feature = "rocm")]
pub mod ffi;
#[cfg(feature = "rocm")]
pub mod rocm;
