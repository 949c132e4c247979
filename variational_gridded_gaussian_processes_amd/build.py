"""Build libvggp_hip.so (gfx950) in-tree with hipcc.  `python -m variational_gridded_gaussian_processes_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["api.hip", "gemm.hip", "factor_build.hip", "chol.hip", "trsm.hip", "eigh.hip", "mspace.hip", "thin.hip", "masked.hip", "comm.hip"]
HEADERS = ["common.h", "ctx.h", "factor_elem.h", "gemm_body.h", os.path.join("..", "..", "include", "vggp.h")]
LIB = os.path.join(HERE, "libvggp_hip.so")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into one shared library (in-tree)."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libvggp_hip.so cannot be built (ROCm toolchain required)")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB + ".tmp"] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)          # a reader (or a snapshot of the tree) never sees a half-written library
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
