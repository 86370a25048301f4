"""In-tree build of the HIP library (gfx950 only).  `python -m pyradiomics_amd._build` or
`__graft_entry__.build()`; the resulting csrc/libpyradiomics_amd.so is git-ignored but travels with
the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpyradiomics_amd.so")
SOURCES = ["prad_api.hip", "prad_firstorder.hip", "prad_features.hip", "prad_resample.hip", "prad_filters.hip"]


def _headers() -> list:
    """every header a translation unit can include: all of csrc/*.h plus the public C ABI"""
    hs = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))
    return hs + [os.path.join("..", "..", "include", "pyradiomics_amd.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + _headers())


def build(force: bool = False, verbose: bool = False) -> str:
    """compiles the translation units concurrently (no device code crosses a unit: no -fgpu-rdc), then links"""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        jobs.append((cmd, obj, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    log = []
    for cmd, obj, proc in jobs:
        out, _ = proc.communicate()
        log.append(" ".join(cmd) + "\n" + out)
        if proc.returncode != 0:
            for _c, _o, other in jobs:
                if other.poll() is None:
                    other.kill()
            raise RuntimeError("hipcc failed:\n%s" % log[-1])
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _c, obj, _p in jobs] + ["-o", LIB + ".tmp"]
    out = subprocess.run(link, cwd=CSRC, capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("hipcc link failed:\n%s\n%s" % (" ".join(link), out.stdout + out.stderr))
    os.replace(LIB + ".tmp", LIB)
    if verbose:
        print("\n".join(log))
        print(" ".join(link))
        print(out.stdout + out.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
