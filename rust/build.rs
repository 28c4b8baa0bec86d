//! build.rs -- compiles the gfx950 attention kernels with hipcc and links them.
//! Replaces the nvcc build of kernels/flash_attn_fwd.cu (reference build.rs:10-70).
//! NOT run in the build image (no cargo there); the same commands are what
//! nano-vllm-rs_amd/build.py runs, and tests/test_rust_crate.py keeps the two source lists equal.
use std::{env, path::PathBuf, process::Command};

/// Sources under NVLLM_CSRC (default: ../nano-vllm-rs_amd/csrc), host C++ and HIP alike: hipcc
/// compiles both with `-x hip`.
const SOURCES: &[&str] = &[
    "nvllm_host.cpp",
    "nvllm_sched.cpp",
    "capi.cpp",
    "launch_policy.cpp",
    "reshape_and_cache.hip",
    "qk_norm_rope_cache.hip",
    "paged_attn.hip",
    "paged_attn_mid.hip",
    "flash_attn.hip",
    "flash_attn_mfma.hip",
];

fn main() {
    if env::var("CARGO_FEATURE_ROCM").is_err() {
        return;
    }
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    if Command::new(&hipcc).arg("--version").output().is_err() {
        // like the reference when nvcc is missing (build.rs:20-25): warn and skip
        println!("cargo:warning=hipcc not found, skipping the gfx950 kernel build");
        return;
    }
    let csrc = PathBuf::from(env::var("NVLLM_CSRC").unwrap_or_else(|_| "../nano-vllm-rs_amd/csrc".into()));
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-env-changed=NVLLM_CSRC");
    let mut objs = Vec::new();
    for s in SOURCES {
        let obj = out.join(format!("{s}.o"));
        let ok = Command::new(&hipcc)
            .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-x", "hip", "-c"])
            .arg(csrc.join(s))
            .arg("-o")
            .arg(&obj)
            .status()
            .map(|st| st.success())
            .unwrap_or(false);
        assert!(ok, "hipcc failed on {s}");
        objs.push(obj);
    }
    let lib = out.join("libnvllm_attn.so");
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"])
        .arg(&lib)
        .args(&objs)
        .status()
        .map(|st| st.success())
        .unwrap_or(false);
    assert!(ok, "linking libnvllm_attn.so failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=nvllm_attn");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}
