// build.rs — compiles the HIP backend for gfx950 with hipcc and links it.
// NOTE: written but never compiled in this repository's build environment (no cargo/rustc there).
// tests/test_rust_sources.py keeps SRCS and FLAGS below equal to path_tracer_amd/csrc/Makefile.
use std::{env, path::PathBuf, process::Command};

const SRCS: &[&str] = &[
    "insploray_hip.hip", "insploray_comm.hip", "pt_kernels.hip", "pt_wavefront.hip", "pt_selftest.hip", "pt_lbvh_build.hip",
    "host_math.cpp", "exr_io.cpp", "exr_codecs.cpp", "lbvh.cpp", "sky_distribution.cpp",
];
// -ffp-contract=off / -fno-fast-math: no FMA contraction, results stay bit-comparable with the CPU restatement of the
// reference; -fno-slp-vectorize: packed f32 ops issue slower than two scalar ones on gfx950
const FLAGS: &[&str] = &["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize"];
const LIBS: &[&str] = &["-lz", "-ldl"];
const HDRS: &[&str] = &["ir_internal.h", "pt_kernels.h", "pt_device.h", "lbvh.h", "exr_codecs.h", "sky_distribution.h", "pt_lbvh_build.h", "Makefile"];

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..");
    let csrc = root.join("path_tracer_amd/csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("libinsploray_hip.so");
    let status = Command::new(&hipcc)
        .arg("--offload-arch=gfx950")
        .args(FLAGS)
        .arg("-shared")
        .arg("-o").arg(&lib)
        .args(SRCS.iter().map(|f| csrc.join(f)))
        .args(LIBS)
        .status()
        .expect("hipcc not found (set HIPCC)");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=insploray_hip");
    for f in SRCS.iter().chain(HDRS) {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", root.join("include/insploray_hip.h").display());
    println!("cargo:rerun-if-env-changed=HIPCC");
}
