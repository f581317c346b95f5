//! Locates (or builds) libceno_hip.so and libceno_prover.so.
//!
//! * `CENO_HIP_LIB_DIR` = a directory that already holds both libraries (what `python -m ceno_amd.build` leaves in
//!   `ceno_amd/`): only link flags are emitted.
//! * otherwise the libraries are built from the sources of this repository with `hipcc --offload-arch=gfx950` (device TUs
//!   under `ceno_amd/csrc/*.hip`) and the host C++ compiler (`ceno_amd/host/*.cpp`), the same commands as `ceno_amd/build.py`.
use std::{env, path::PathBuf, process::Command};

fn main() {
    println!("cargo:rerun-if-env-changed=CENO_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    let repo = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../..").canonicalize().unwrap();
    let lib_dir = match env::var("CENO_HIP_LIB_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => build_from_source(&repo),
    };
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=ceno_hip");
    println!("cargo:rustc-link-lib=dylib=ceno_prover");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    println!("cargo:include={}", repo.join("include").display());
}

fn build_from_source(repo: &PathBuf) -> PathBuf {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    let hipcc = format!("{}/bin/hipcc", rocm);
    let include = repo.join("include");
    let csrc = repo.join("ceno_amd/csrc");
    let host = repo.join("ceno_amd/host");
    for dir in [&include, &csrc, &host] {
        println!("cargo:rerun-if-changed={}", dir.display());
    }
    let mut objs = vec![];
    for entry in std::fs::read_dir(&csrc).expect("ceno_amd/csrc") {
        let p = entry.unwrap().path();
        if p.extension().map_or(false, |e| e == "hip") {
            let o = out.join(format!("{}.o", p.file_name().unwrap().to_string_lossy()));
            run(Command::new(&hipcc).args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I"]).arg(&include).arg("-c").arg(&p).arg("-o").arg(&o));
            objs.push(o);
        }
    }
    run(Command::new(&hipcc).args(["--offload-arch=gfx950", "-shared", "-fPIC", "-o"]).arg(out.join("libceno_hip.so")).args(&objs));
    let mut cpp = vec![];
    for entry in std::fs::read_dir(&host).expect("ceno_amd/host") {
        let p = entry.unwrap().path();
        if p.extension().map_or(false, |e| e == "cpp") {
            cpp.push(p);
        }
    }
    run(Command::new(env::var("CXX").unwrap_or_else(|_| "g++".into()))
        .args(["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-shared", "-I"]).arg(&include).arg("-I").arg(format!("{}/include", rocm))
        .arg("-o").arg(out.join("libceno_prover.so")).args(&cpp)
        .arg("-L").arg(&out).arg("-lceno_hip").arg("-L").arg(format!("{}/lib", rocm)).args(["-lamdhip64", "-lpthread", "-ldl", "-Wl,-rpath,$ORIGIN"]));
    out
}

fn run(cmd: &mut Command) {
    let status = cmd.status().unwrap_or_else(|e| panic!("failed to start {:?}: {}", cmd, e));
    assert!(status.success(), "{:?} failed", cmd);
}
