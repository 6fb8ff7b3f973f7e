"""Build libmsf2.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "msf2_api.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("msf2_common.h", "msf2_gemm.h", "msf2_gemm_bf16.h", "msf2_gemm_bf16_astat.h", "msf2_attn_fused.h", "msf2_linkv_bf16.h", "msf2_ops.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "msf2.h")]
OUT = os.path.join(HERE, "lib", "libmsf2.so")


STAMP = OUT + ".sources.sha256"


def sources_digest() -> str:
    """SHA-256 over the sources the library is built from (and the flags): the library is rebuilt whenever this
    differs from the digest recorded beside it, whatever the files' time stamps say (a fresh clone, a checkout)."""
    h = hashlib.sha256()
    for d in DEPS:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    h.update(os.environ.get("MSF2_CXXFLAGS", "").encode())
    return h.hexdigest()


def up_to_date() -> bool:
    try:
        with open(STAMP) as f:
            return os.path.exists(OUT) and f.read().strip() == sources_digest()
    except OSError:
        return False


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MSF2_CXXFLAGS", "").split()
    cmd = [hipcc, "-O3", *extra, "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-Wno-inline-asm", SRC, "-o", OUT]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as f:
        f.write(sources_digest() + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
