#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/g.py <<PY
import sys; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine
for n, dist in ((256, "smooth"), (512, "smooth"), (512, "uniform")):
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
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/glszm_r03 -o g -- python /tmp/g.py 2>&1 | grep glszm_compact
python $R/scripts/rocpd_stats.py $R/gpurun_out/glszm_r03/g_results.db | head -30
find $R/gpurun_out/glszm_r03 -name "*.db" -delete
