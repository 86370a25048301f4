"""In-tree build of the HIP library (gfx950 only).  `python -m pyradiomics_amd._build` or
`__graft_entry__.build()`; the resulting csrc/libpyradiomics_amd.so is git-ignored but travels with
the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpyradiomics_amd.so")
SOURCES = ["prad_api.hip", "prad_firstorder.hip", "prad_features.hip", "prad_resample.hip"]
HEADERS = ["prad_runtime.h", "kernels_generic.h", "kernels_sweep.h", "kernels_neigh.h", "kernels_glszm.h", "kernels_filters.h", "kernels_voxel.h", "kernels_voxtex.h", "kernels_binning.h", "kernels_firstorder.h", "kernels_features.h", "kernels_resample.h",
           os.path.join("..", "..", "include", "pyradiomics_amd.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function"] + SOURCES + ["-o", LIB + ".tmp"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), out.stdout + out.stderr))
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        print(" ".join(cmd))
        print(out.stdout + out.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
