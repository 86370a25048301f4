#!/bin/bash
# round 4: what the unions of glszm_pairs_kernel are made of (build_variants/lib_pstats.so, -DPRAD_PAIRS_STATS)
R=$GRAFT_REPO_ROOT
cat > /tmp/gs.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch
from bench import make_volume
from pyradiomics_amd import engine
for n, dist in ((256, "smooth"), (512, "smooth"), (512, "uniform")):
    img, msk = make_volume(n, 32, dist, 0, torch.device("cuda", 0))
    engine.glszm_compact(img, msk, 32, img.numel())
    torch.cuda.synchronize()
    print("==", n, dist, flush=True)
PY
PRAD_LIB=$R/build_variants/lib_pstats.so python /tmp/gs.py 2>&1 | grep -v amdgpu.ids
