"""Builds libceno_hip.so (HIP kernels + C ABI) and libceno_prover.so (C++ host layer) in-tree.

hipcc cross-compiles for gfx950 without a GPU; objects are cached under ceno_amd/_build and only
stale translation units are recompiled.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
OBJ = os.path.join(HERE, "_build")
LIB_HIP = os.path.join(HERE, "libceno_hip.so")
LIB_PROVER = os.path.join(HERE, "libceno_prover.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# HOST_TUNE: scheduling model for the host halves (no instruction-set extension, the binaries run on any x86-64).  clang's generic
# x86-64 model turns the branch-free 128-bit reductions of the host Poseidon2 (csrc/poseidon2_host.hpp: the Fiat-Shamir challenger
# and the host-finished tree tops) into code that runs 2.8x slower: 1.55 vs 0.54 us per permutation on the GPU box's EPYC 9575F.
HOST_TUNE = "-mtune=znver3"
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-fno-gpu-rdc", "-Xarch_host", HOST_TUNE, "-I", os.path.join(ROOT, "include")] + os.environ.get("CENO_HIP_EXTRA_FLAGS", "").split()
CXX_FLAGS = ["-O3", HOST_TUNE, "-std=c++17", "-fPIC", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
             "-D__HIP_PLATFORM_AMD__"]
HOST_CXX = "/opt/rocm/lib/llvm/bin/clang++"  # the ROCm clang (0.54 us per host permutation; g++ -O3: 0.60, g++ -O2: 0.68)


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def _flags_changed() -> bool:
    """objects are only reusable when they were compiled with the same flags (tools/ab_*.sh rebuild with CENO_HIP_EXTRA_FLAGS):
    the flag string is part of the cache key, so a variant build never masquerades as the shipped one"""
    stamp = os.path.join(OBJ, "flags.txt")
    cur = " ".join(os.environ.get("CENO_HIP_EXTRA_FLAGS", "").split())  # (the fixed flags hold checkout-dependent paths)
    old = open(stamp).read() if os.path.exists(stamp) else ""
    if old != cur:
        with open(stamp, "w") as f:
            f.write(cur)
        return True
    return False


def device_asm_path(src: str) -> str:
    """gfx950 assembly of one .hip translation unit, as assembled into its object (kept by build_hip)"""
    return os.path.join(OBJ, os.path.basename(src) + ".gfx950.s")


def _keep_device_asm(src: str):
    """after a -save-temps=obj compile: keep `<stem>-hip-amdgcn-amd-amdhsa-gfx950.s` as `<name>.hip.gfx950.s`, drop the other temporaries"""
    stem = os.path.splitext(os.path.basename(src))[0]
    dev = os.path.join(OBJ, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    if os.path.exists(dev):
        os.replace(dev, device_asm_path(src))
    for f in glob.glob(os.path.join(OBJ, stem + "-hip-amdgcn-amd-amdhsa-gfx950.*")) + glob.glob(os.path.join(OBJ, stem + "-host-x86_64-unknown-linux-gnu.*")) + \
            glob.glob(os.path.join(OBJ, os.path.basename(src) + "-hip-amdgcn-amd-amdhsa.hipfb")):
        try:
            os.remove(f)
        except OSError:
            pass


def device_asm_files(build: bool = True):
    """[(source, path of its gfx950 assembly)] for every .hip unit; builds what is missing or stale first"""
    if build:
        build_hip()
    return [(s, device_asm_path(s)) for s in sorted(glob.glob(os.path.join(CSRC, "*.hip")))]


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    force = force or _flags_changed()
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h"))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs) or not os.path.exists(device_asm_path(s)):
            # -save-temps=obj: the device assembly that is ASSEMBLED INTO this object is kept beside it (no second compile) — what
            # tests/test_isa_lint.py reads: hand-written wait states inside `asm` statements are invisible to the compiler's hazard recogniser
            jobs.append(([HIPCC] + HIP_FLAGS + ["-save-temps=obj", "-c", s, "-o", o], s))
    if jobs:
        def compile_one(job):
            cmd, src = job
            out = _run(cmd)
            _keep_device_asm(src)
            return out

        with cf.ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as ex:
            for out in ex.map(compile_one, jobs):
                if verbose and out.strip():
                    print(out)
    if force or jobs or _stale(LIB_HIP, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs)
    return LIB_HIP


def build_prover(force: bool = False) -> str:
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    if not srcs:
        return ""
    hdrs = glob.glob(os.path.join(HOST, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if force or _stale(LIB_PROVER, srcs + hdrs + [LIB_HIP]):
        cxx = HOST_CXX if os.path.exists(HOST_CXX) else (shutil.which("g++") or "g++")
        _run([cxx] + CXX_FLAGS + ["-shared", "-o", LIB_PROVER] + srcs +
             ["-L", HERE, "-lceno_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib",
              "-lpthread", "-ldl"])
    return LIB_PROVER


def build_all(force: bool = False, verbose: bool = False):
    build_hip(force, verbose)
    build_prover(force)


SAN = os.path.join(HERE, "_san")


def build_sanitized(verbose: bool = True) -> dict:
    """CPU-ONLY sanitizer builds (never the GPU code, never shipped; verdict round 4 item 7):
      * oracle/_san/libceno_oracle_asan.so     the C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (gcc)
      * ceno_amd/_san/libceno_prover_asan.so   the C++ host layer (host/*.cpp: transcript, virtual polynomials, host arithmetic, the exchange,
                                               the proof drivers) under ASan + UBSan (g++), linked against the ordinary libceno_hip.so
      * ceno_amd/_san/exchange_stress_tsan / _asan   tools/sanitize/exchange_stress.cpp + host/dist.cpp + transcript.cpp under
                                               ThreadSanitizer / ASan: the shared-memory exchange between `world` threads-as-ranks and the
                                               pool's spin lock
    tests/test_sanitizers.py (CENO_RUN_SANITIZERS=1) runs the CPU test-suite subset against the first two (LD_PRELOAD of the ASan runtime,
    CENO_PROVER_LIB / CENO_ORACLE_LIB) and the drivers; profiles/r05_sanitizers.log is such a run."""
    build_all()
    os.makedirs(SAN, exist_ok=True)
    osan = os.path.join(ROOT, "oracle", "_san")
    os.makedirs(osan, exist_ok=True)
    gcc, gxx = shutil.which("gcc") or "gcc", shutil.which("g++") or "g++"
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    out = {}
    osrcs = [os.path.join(ROOT, "oracle", f) for f in ("oracle.c", "tower.c", "commit.c", "rotation.c", "basefold.c", "witgen.c", "transcript.c", "dense_avx512.c")]
    out["oracle_asan"] = os.path.join(osan, "libceno_oracle_asan.so")
    _run([gcc] + san + ["-march=x86-64-v3", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", out["oracle_asan"]] + osrcs)
    hsrcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    link = ["-L", HERE, "-lceno_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-ldl"]
    inc = ["-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]
    out["prover_asan"] = os.path.join(SAN, "libceno_prover_asan.so")
    _run([gxx] + san + inc + ["-shared", "-o", out["prover_asan"]] + hsrcs + link)
    drv = [os.path.join(ROOT, "tools", "sanitize", "exchange_stress.cpp"), os.path.join(HOST, "dist.cpp"), os.path.join(HOST, "transcript.cpp")]
    stub = os.path.join(SAN, "stub.cpp")  # dist.cpp's references into the other host units (not on the exchange path)
    with open(stub, "w") as f:
        f.write('#include <string>\nstatic thread_local std::string e; int prover_set_error(int c, const char* m) { e = m ? m : ""; return c; }\n'
                'extern "C" const char* ceno_prover_last_error(void) { return e.c_str(); }\n')
    out["exchange_tsan"] = os.path.join(SAN, "exchange_stress_tsan")
    out["exchange_asan"] = os.path.join(SAN, "exchange_stress_asan")
    for key, flags in (("exchange_tsan", ["-fsanitize=thread", "-g", "-O1"]), ("exchange_asan", san)):
        try:
            _run([gxx] + flags + inc + ["-o", out[key]] + drv + [stub] + link + ["-lrt"])
        except RuntimeError as e:
            if "undefined reference" not in str(e):
                raise
            # dist.cpp calls into more host units than the stub covers: link the whole host layer (same flags)
            _run([gxx] + flags + inc + ["-o", out[key], drv[0]] + hsrcs + link + ["-lrt"])
    out["asan_runtime"] = subprocess.run([gcc, "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    out["ubsan_runtime"] = subprocess.run([gcc, "-print-file-name=libubsan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if verbose:
        for k, v in out.items():
            print("sanitize:", k, v)
    return out


if __name__ == "__main__":
    if "--sanitize" in sys.argv:
        build_sanitized()
        sys.exit(0)
    build_all(force="--force" in sys.argv, verbose=True)
    print("built", LIB_HIP, LIB_PROVER if os.path.exists(LIB_PROVER) else "")
