#!/bin/bash
# usage: prof_glszm.sh <tag> [N] [dist] -- rocprofv3 kernel stats of 5 GLSZM builds on a synthetic N^3 volume
tag=$1; N=${2:-256}; dist=${3:-smooth}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/glszm_run.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch
from bench import make_volume
from pyradiomics_amd import engine
img, mask = make_volume($N, 32, "$dist", 0, torch.device("cuda", 0))
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    P, sizes = engine.glszm_compact(img, mask, 32)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("glszm_compact $N^3 $dist: %.2f ms wall, %d distinct sizes, max %d, zones %d" % (dt * 1e3, len(sizes), sizes.max(), int(P.sum().item())))
PY
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/glszm_$tag -o $tag -- python /tmp/glszm_run.py > $R/gpurun_out/glszm_$tag.log 2>&1
tail -2 $R/gpurun_out/glszm_$tag.log
python $R/scripts/rocpd_stats.py $R/gpurun_out/glszm_$tag/${tag}_results.db | head -24
