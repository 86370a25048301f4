#!/bin/bash
# round 4: how much of glszm_border8s_kernel is the union phase?  (build_variants/lib_nounion.so: -DPRAD_DBG_NOUNION, wrong zones)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/g.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine
n, dist = int(os.environ["GN"]), os.environ["GD"]
img, msk = make_volume(n, 32, dist, 0, torch.device("cuda", 0))
for _ in range(6):
    try: engine.glszm_compact(img, msk, 32, img.numel())
    except Exception as e: pass
torch.cuda.synchronize()
PY
for lib in "" "$R/build_variants/lib_nounion.so"; do
for c in "256 smooth" "512 smooth"; do
  set -- $c
  rm -rf /tmp/gz
  echo "== ${lib:-default} $c"
  PRAD_LIB=$lib GN=$1 GD=$2 rocprofv3 --kernel-trace --stats -d /tmp/gz -o g -- python /tmp/g.py > /dev/null 2>&1
  python $R/scripts/rocpd_stats.py /tmp/gz/g_results.db | grep -E "glszm_(border8s|tile8|rootsum)"
done; done
