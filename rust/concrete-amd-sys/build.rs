//! Compiles the HIP sources into `libconcrete_amd.so` with hipcc for gfx950 and tells cargo to link
//! it. The command line is the one `__graft_entry__.build()` of the repository uses.
use std::env;
use std::path::PathBuf;
use std::process::Command;

const SOURCES: [&str; 9] = [
    "abi.hip",
    "keyswitch.hip",
    "keyswitch_mfma.hip",
    "linear.hip",
    "pbs_generic.hip",
    "pbs_block.hip",
    "pbs_wave.hip",
    "pbs_wave512.hip",
    "probe.hip",
    "client.cpp",
];

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let repo = manifest.join("../..");
    let csrc = repo.join("concrete_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let lib = out.join("libconcrete_amd.so");

    let status = Command::new(&hipcc)
        .current_dir(&csrc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip"])
        .args(SOURCES.iter().map(|f| csrc.join(f)))
        .arg("-o")
        .arg(&lib)
        .arg("-lpthread")
        .status()
        .unwrap_or_else(|e| panic!("could not run {hipcc}: {e}"));
    assert!(status.success(), "hipcc failed to build libconcrete_amd.so");

    for f in SOURCES {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", csrc.join("common.hpp").display());
    println!("cargo:rerun-if-changed={}", csrc.join("kernels.hpp").display());
    println!("cargo:rerun-if-changed={}", repo.join("include/concrete_amd.h").display());
    println!("cargo:rerun-if-changed={}", repo.join("include/concrete_amd_client.h").display());
    println!("cargo:rerun-if-env-changed=HIPCC");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=concrete_amd");
    // dependants find the headers here (DEP_CONCRETE_AMD_INCLUDE)
    println!("cargo:include={}", repo.join("include").display());
}
