#!/bin/bash
# round 4: wavelet (8 coif1 sub-bands of a 256^3 volume): fused 3-axis kernel vs the three axis passes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/w1.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine, _lib
size = int(os.environ.get("WSIZE", "256"))
lv, msk = make_volume(size, 32, "smooth", 0, torch.device("cuda", 0))
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
engine.wavelet_images(img); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): engine.wavelet_images(img)
torch.cuda.synchronize()
print("fused=%s size=%d path=%s: %.3f ms per 8 sub-bands" % (os.environ.get("PRAD_SWT_NOFUSE") is None, size, _lib.last_path(), (time.perf_counter() - t0) / 5 * 1e3), flush=True)
PY
for m in "" "PRAD_SWT_NOFUSE=1" "WSIZE=512" "WSIZE=512 PRAD_SWT_NOFUSE=1"; do
  rm -rf /tmp/w1
  env $m rocprofv3 --kernel-trace --stats -d /tmp/w1 -o g -- python /tmp/w1.py 2>&1 | grep "fused="
  python $R/scripts/rocpd_stats.py /tmp/w1/g_results.db | grep -E "swt|elementwise|kernel \||---" | head -6
done
