#!/bin/bash
# round 4: GLSZM kernels per volume (256^3 uniform / smooth, 512^3 smooth), each in its own profile
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/g.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine
n, dist = int(os.environ["GN"]), os.environ["GD"]
img, msk = make_volume(n, 32, dist, 0, torch.device("cuda", 0))
for _ in range(3):
    engine.glszm_compact(img, msk, 32, img.numel())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    engine.glszm_compact(img, msk, 32, img.numel())
torch.cuda.synchronize()
print("glszm_compact %d %s: %.3f ms wall" % (n, dist, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
PY
for c in "256 uniform" "256 smooth" "512 smooth" "512 uniform"; do
  set -- $c
  rm -rf /tmp/gz
  GN=$1 GD=$2 rocprofv3 --kernel-trace --stats -d /tmp/gz -o g -- python /tmp/g.py 2>&1 | grep glszm_compact
  python $R/scripts/rocpd_stats.py /tmp/gz/g_results.db | grep -v "at::\|elementwise\|Memcpy\|fill" | head -14
done
