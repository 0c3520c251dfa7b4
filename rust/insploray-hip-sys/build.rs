// build.rs — compiles the HIP backend for gfx950 with hipcc and links it.
// NOTE: written but never compiled in this repository's build environment (no cargo/rustc there).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..");
    let csrc = root.join("path_tracer_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("libinsploray_hip.so");
    let status = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"])
        // no FMA contraction: results stay bit-comparable with the CPU restatement of the reference
        .args(["-ffp-contract=off", "-fno-fast-math"])
        .arg("-o").arg(&lib)
        .args(["insploray_hip.hip", "pt_kernels.hip", "pt_wavefront.hip", "pt_selftest.hip", "host_math.cpp", "exr_io.cpp", "exr_codecs.cpp", "lbvh.cpp"].map(|f| csrc.join(f)))
        .arg("-lz")
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=insploray_hip");
    for f in ["insploray_hip.hip", "pt_kernels.hip", "pt_wavefront.hip", "pt_selftest.hip", "pt_kernels.h", "pt_device.h", "lbvh.h", "lbvh.cpp", "host_math.cpp", "exr_io.cpp", "exr_codecs.cpp", "exr_codecs.h"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
}
