"""Build libmsf2.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "msf2_api.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("msf2_common.h", "msf2_gemm.h", "msf2_gemm_bf16.h", "msf2_ops.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "msf2.h")]
OUT = os.path.join(HERE, "lib", "libmsf2.so")


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if os.environ.get("MSF2_CXXFLAGS"):
        force = True                 # diagnostic builds never reuse the product library
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MSF2_CXXFLAGS", "").split()
    cmd = [hipcc, "-O3", *extra, "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-inline-asm", SRC, "-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
