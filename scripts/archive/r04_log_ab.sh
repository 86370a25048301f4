#!/bin/bash
# round 4: A/B of the LoG pass kernel variants in build_variants/ (scripts/build_filter_variant.sh), 256^3, five sigmas
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/l1.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine, _lib
lv, msk = make_volume(256, 32, "smooth", 0, torch.device("cuda", 0))
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
sig = (1.0, 2.0, 3.0, 4.0, 5.0)
def run(): engine.log_images(img, (1.0, 1.0, 1.0), sig)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): run()
torch.cuda.synchronize()
print("%s: %.3f ms per 5 sigmas" % (os.path.basename(os.environ.get("PRAD_LIB", "default")), (time.perf_counter() - t0) / 4 * 1e3), flush=True)
PY
for lib in $R/build_variants/lib_*.so; do
  rm -rf /tmp/l1
  PRAD_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/l1 -o g -- python /tmp/l1.py 2>&1 | grep "ms per 5"
  python $R/scripts/rocpd_stats.py /tmp/l1/g_results.db | grep -E "rgauss"
done
