"""Build libssq_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m ssqueeze_rs_amd.build` or `build_lib()`.  hipcc cross-compiles without a GPU.
The built .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libssq_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"

SOURCES = ["api_common.hip", "api_stft.hip", "stft_fused.hip", "stft_anylen.hip", "stft_generic.hip",
           "api_cwt.hip", "cwt_kernels.hip", "cwt_reg.hip", "cwt_os.hip", "frontend.hip", "fft_generic.hip", "api_icwt.hip", "host_cache.hip", "api_upstream.hip", "api_gather.hip"]
CXXFLAGS = ["-std=c++17", "-O3", "-fno-slp-vectorize", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-unused-value"]


def _hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libssq_hip.so cannot be built")


def _newer(src_list, target) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_lib(force: bool = False, verbose: bool = True, extra_flags=(), suffix: str = "") -> str:
    """suffix/extra_flags build a diagnostic variant (e.g. -DSSQ_STAMPS) next to the product library."""
    global OBJ_DIR, LIB
    lib = LIB if not suffix else LIB.replace(".so", f"_{suffix}.so")
    obj_dir = OBJ_DIR if not suffix else OBJ_DIR + "_" + suffix
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "ssq_hip.h"))
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer([s] + headers, o):
            jobs.append([hipcc, *CXXFLAGS, *extra_flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(objs, lib):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_lib(extra_flags=("-DSSQ_STAMPS", "-DSSQ_ABLATE_HOOKS", "-DSSQ_TUNING"), suffix="diag"))
    elif "--abl" in sys.argv:       # timing experiments: SSQ_ABLATE=<mask> skips stages (results wrong)
        print(build_lib(extra_flags=("-DSSQ_ABLATE_HOOKS", "-DSSQ_TUNING"), suffix="abl"))
    elif "--tune" in sys.argv:      # the measured-slower alternatives behind their environment switches (SSQ_CWT_FUSED, ...)
        print(build_lib(extra_flags=("-DSSQ_TUNING",), suffix="tune"))
    elif "--variant" in sys.argv:   # python -m ssqueeze_rs_amd.build --variant NAME -DFOO=1 ...  -> libssq_hip_NAME.so
        name = sys.argv[sys.argv.index("--variant") + 1]
        idx = sys.argv.index("--variant") + 2          # everything after the name goes to hipcc (-D..., -mllvm ...)
        print(build_lib(extra_flags=tuple(sys.argv[idx:]), suffix=name))
    else:
        print(build_lib(force="--force" in sys.argv))
