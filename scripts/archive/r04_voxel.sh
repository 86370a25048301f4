#!/bin/bash
# round 4: voxel maps at 512^3 (bench.py's modes.voxel / voxel3d): sliding-window kernel vs the from-scratch window kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/v1.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
import bench
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
for three_d in (False, True):
    nk, dt, kms = bench.mode_voxel(dev, 0, 1, int(os.environ.get("VSIZE", "512")), torch.cuda.synchronize, three_d)
    print("slide=%s 3d=%s: %.1f M kernels/s (%.2f ms device, variant %s)" % (os.environ.get("PRAD_VOX_NO_SLIDE") is None, three_d, nk / dt / 1e6, kms, engine.last_variant()), flush=True)
PY
for m in "" "PRAD_VOX_NO_SLIDE=1"; do
  rm -rf /tmp/v1
  env $m rocprofv3 --kernel-trace --stats -d /tmp/v1 -o g -- python /tmp/v1.py 2>&1 | grep "slide="
  python $R/scripts/rocpd_stats.py /tmp/v1/g_results.db | grep -E "voxel|kernel \||---" | head -8
done
