#!/bin/bash
# round 4: where a workgroup of glszm_tile8_kernel spends its cycles, phase by phase (build_variants/lib_t8prof.so, -DPRAD_T8_PROF)
R=$GRAFT_REPO_ROOT
cat > /tmp/gp.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch
from bench import make_volume
from pyradiomics_amd import engine
for n, dist in ((512, "smooth"), (512, "uniform")):
    img, msk = make_volume(n, 32, dist, 0, torch.device("cuda", 0))
    engine.glszm_compact(img, msk, 32, img.numel())
    torch.cuda.synchronize()
    print("==", n, dist, flush=True)
PY
PRAD_LIB=$R/build_variants/lib_t8prof.so python /tmp/gp.py 2>&1 | grep -v amdgpu.ids | tr "\n" " "
