#!/bin/bash
# round 4: what bounds glszm_tile8_kernel?  ablation builds (build_variants/lib_t8a.so: -DPRAD_DBG_T8=1 no in-tile unions,
# lib_t8b.so: =2 no unions and no finds; both give wrong zones) + PMC counters of the default build
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_t8
rm -rf $O; mkdir -p $O
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
for lib in "" "$R/build_variants/lib_t8a.so" "$R/build_variants/lib_t8b.so"; do
for c in "256 uniform" "256 smooth" "512 smooth"; do
  set -- $c
  rm -rf /tmp/gz
  echo "== ${lib:-default} $c"
  PRAD_LIB=$lib GN=$1 GD=$2 rocprofv3 --kernel-trace --stats -d /tmp/gz -o g -- python /tmp/g.py > /dev/null 2>&1
  python $R/scripts/rocpd_stats.py /tmp/gz/g_results.db | grep -E "glszm_" | head -8
done; done 2>&1 | tee $O/ablate.txt
pass() { name=$1; shift; GN=512 GD=smooth rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python /tmp/g.py > $O/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $O/pmc > $O/pmc.md
find $O -name "*.csv" -delete
awk '/^### /{p=0} /glszm_tile8|glszm_border8s|glszm_rootsum/{p=1} p' $O/pmc.md
