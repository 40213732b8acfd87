// build.rs -- links the MI355X backend (libapexgpu.so, built by apex-solver_amd/csrc/Makefile with hipcc).
// Set APEXGPU_LIB_DIR to the directory that holds libapexgpu.so; the HIP runtime and RCCL come from ROCM_PATH.
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=APEXGPU_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    if env::var("CARGO_FEATURE_GPU").is_err() {
        return; // the `gpu` feature gates every item of src/linearizer/gpu and src/linalg/gpu_schur.rs
    }
    let lib_dir = env::var("APEXGPU_LIB_DIR").expect("APEXGPU_LIB_DIR must point at the directory of libapexgpu.so");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={lib_dir}");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=apexgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{lib_dir}");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{rocm}/lib");
}
