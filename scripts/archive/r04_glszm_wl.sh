#!/bin/bash
# round 4: GLSZM with the cross-tile work list (default) against the border scans (PRAD_GLSZM_BORDER=scan) and the
# path-halving probe build (build_variants/lib_t8halve.so): parity tests first, then per-kernel device times
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_glszm_wl
rm -rf $O; mkdir -p $O
cd $R
[ -z "$NOTESTS" ] && timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sharded.py tests/test_gpu_features.py -x -q -m gpu -k "glszm or zone or fuzz or sharded" 2>&1 | tail -5 | tee $O/tests.txt
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
    P, sizes = engine.glszm_compact(img, msk, 32, img.numel())
torch.cuda.synchronize()
print("glszm_compact %d %s: %.3f ms wall  zones %d  distinct sizes %d  checksum %.0f" % (n, dist, (time.perf_counter() - t0) / 5 * 1e3, int(P.sum().item()), len(sizes), float((P.sum(0).cpu().numpy() * sizes).sum())), flush=True)
PY
for V in ${VARIANTS:-dense}; do
  lib=""; bor=""
  [ $V != dense ] && lib=$R/build_variants/lib_$V.so
  for c in ${CASES:-256_uniform 256_smooth 512_smooth 512_uniform}; do
    set -- ${c/_/ }
    rm -rf /tmp/gz
    echo "== $V $c"
    PRAD_LIB=$lib PRAD_GLSZM_BORDER=$bor GN=$1 GD=$2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/gz -o g -- python /tmp/g.py 2>&1 | grep glszm_compact
    python $R/scripts/rocpd_stats.py /tmp/gz/g_results.db | grep -E "glszm_|pack_levels|neigh_pack" | head -12
  done
done 2>&1 | tee $O/kernels.txt
