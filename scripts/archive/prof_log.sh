#!/bin/bash
# LoG kernels at 256^3: one sigma per launch sequence vs five (prad_log_multi_dev)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/l1.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch, time
from bench import make_volume
from pyradiomics_amd import engine
lv, msk = make_volume(256, 32, "smooth", 0, torch.device("cuda", 0))
img = (lv.to(torch.float32) * 25.0 + 3.0).to(torch.int16)
sig = (1.0, 2.0, 3.0, 4.0, 5.0)
multi = os.environ.get("LOG_MULTI", "1") == "1"
def run():
    if multi: engine.log_images(img, (1.0, 1.0, 1.0), sig)
    else:
        for s in sig: engine.log_image(img, (1.0, 1.0, 1.0), s)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4): run()
torch.cuda.synchronize()
print("multi=%s: %.3f ms per 5 sigmas" % (multi, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
PY
for m in 0 1; do
  rm -rf /tmp/l1
  LOG_MULTI=$m rocprofv3 --kernel-trace --stats -d /tmp/l1 -o g -- python /tmp/l1.py 2>&1 | grep "multi="
  python $R/scripts/rocpd_stats.py /tmp/l1/g_results.db | grep -E "rgauss|kernel \||---"
done
