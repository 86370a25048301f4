#!/bin/bash
# round 4: PMC counters of the dense-model GLSZM kernels (512^3 smooth by default)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_glszm_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cat > /tmp/g.py <<PY
import sys, os; sys.path.insert(0, "$R")
import torch
from bench import make_volume
from pyradiomics_amd import engine
n, dist = int(os.environ.get("GN", 512)), os.environ.get("GD", "smooth")
img, msk = make_volume(n, 32, dist, 0, torch.device("cuda", 0))
for _ in range(3):
    engine.glszm_compact(img, msk, 32, img.numel())
torch.cuda.synchronize()
PY
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python /tmp/g.py > $O/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT
pass sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $O/pmc > $O/pmc.md
find $O -name "*.csv" -delete; rm -rf $O/pmc
awk '/^### /{p=0} /glszm_tile8|glszm_pairs/{p=1} p' $O/pmc.md
