#!/bin/bash
# round 4: LoG at 256^3 (five sigmas per launch sequence): LDS tile kernels vs the round-3 two-sweep kernels, per-kernel stats
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/l1.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine, _lib
size = int(os.environ.get("LOG_SIZE", "256"))
lv, msk = make_volume(size, 32, "smooth", 0, torch.device("cuda", 0))
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
if os.environ.get("LOG_F64"): img = img.to(torch.float64)
sig = (1.0, 2.0, 3.0, 4.0, 5.0)
def run(): engine.log_images(img, (1.0, 1.0, 1.0), sig)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): run()
torch.cuda.synchronize()
print("tile=%s f64=%s size=%d path=%s: %.3f ms per 5 sigmas" % (os.environ.get("PRAD_LOG_OLDLINE") is None, bool(os.environ.get("LOG_F64")), size, _lib.last_path(), (time.perf_counter() - t0) / 4 * 1e3), flush=True)
PY
for m in "" "PRAD_LOG_PLAIN=1" "LOG_F64=1" "LOG_SIZE=512"; do
  rm -rf /tmp/l1
  env $m rocprofv3 --kernel-trace --stats -d /tmp/l1 -o g -- python /tmp/l1.py 2>&1 | grep "tile="
  python $R/scripts/rocpd_stats.py /tmp/l1/g_results.db | grep -E "rgauss|combine|kernel \||---"
done
