#!/bin/bash
# usage: prof_glszm_lll.sh <tag> [N] -- GLSZM kernels on the wavelet-LLL band (very smooth: giant zones) of the smooth
# volume inside a ball ROI cropped to its bounding box (the last derived image of scripts/bench_cases.py)
tag=$1; N=${2:-256}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/glszm_lll.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch
from bench import make_volume
from pyradiomics_amd import engine
dev = torch.device("cuda", 0)
N = $N
img, _ = make_volume(N, 32, "smooth", 0, dev)
z = torch.arange(N, device=dev) - N / 2
ball = ((z[:, None, None] ** 2 + z[None, :, None] ** 2 + z[None, None, :] ** 2) < (0.45 * N) ** 2)
lll = engine.wavelet_images((img * 25).to(torch.int16))["wavelet-LLL"]
idx = torch.nonzero(ball.any(dim=2).any(dim=1)).flatten()
lo, hi = int(idx[0]), int(idx[-1]) + 1
lll = lll[lo:hi, lo:hi, lo:hi].contiguous(); roi = ball[lo:hi, lo:hi, lo:hi].contiguous()
lv, Ng, _ = engine.bin_image(lll, roi, binCount=32)
for i in range(4):
    torch.cuda.synchronize(); t = time.perf_counter()
    P, sizes = engine.glszm_compact(lv, roi, Ng)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("glszm_compact LLL ball %s: %.2f ms wall, %d distinct sizes, max %d, zones %d" % (tuple(lv.shape), dt * 1e3, len(sizes), sizes.max(), int(P.sum().item())))
PY
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/glszm_$tag -o $tag -- python /tmp/glszm_lll.py > $R/gpurun_out/glszm_$tag.log 2>&1
grep "glszm_compact" $R/gpurun_out/glszm_$tag.log
python $R/scripts/rocpd_stats.py $R/gpurun_out/glszm_$tag/${tag}_results.db | grep glszm
