#!/bin/bash
# per-kernel device times of GLSZM (zones + ranked sizes + compact fill + formulas) for one volume: size dist
R=$GRAFT_REPO_ROOT
N=${1:-512}; D=${2:-smooth}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/g1.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine
img, msk = make_volume($N, 32, "$D", 0, torch.device("cuda", 0))
for _ in range(3):
    engine.glszm_features(img, msk, 32, img.numel())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    engine.glszm_features(img, msk, 32, img.numel())
torch.cuda.synchronize()
print("glszm_features $N $D: %.3f ms wall" % ((time.perf_counter() - t0) / 8 * 1e3), flush=True)
PY
rm -rf /tmp/g1
rocprofv3 --kernel-trace --stats -d /tmp/g1 -o g -- python /tmp/g1.py 2>&1 | grep glszm_features
python $R/scripts/rocpd_stats.py /tmp/g1/g_results.db | grep -E "prad|kernel \||---|rocclr"
