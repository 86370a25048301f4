cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/fo_run.py <<PY
import sys, time
sys.path.insert(0, "$R")
import torch
from pyradiomics_amd import engine
N=512
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
img = (torch.randn((N, N, N), generator=g, device=dev) * 300 + 800).to(torch.int16)
mask = torch.ones((N, N, N), dtype=torch.uint8, device=dev)
for _ in range(2):
    engine.firstorder_stats(img, mask, 0.0)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(6):
    st = engine.firstorder_stats(img, mask, 0.0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 6
print("segment %d^3 int16: %.3f ms wall, kernels %.3f ms" % (N, dt * 1e3, engine.last_kernel_ms("firstorder")))
PY
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fo_prof -o fo -- python /tmp/fo_run.py > $R/gpurun_out/fo_prof.log 2>&1
grep segment $R/gpurun_out/fo_prof.log
python $R/scripts/rocpd_stats.py $R/gpurun_out/fo_prof/fo_results.db | grep "prad::" | head -12
