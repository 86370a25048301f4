#!/bin/bash
# PMC counter passes for the GLSZM kernels (separate passes, kernel trace only).  usage: prof_pmc_glszm.sh <tag> [N] [dist]
tag=$1; N=${2:-512}; dist=${3:-smooth}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_glszm_$tag
mkdir -p $out
cat > /tmp/glszm_pmc_run.py <<PY
import sys
sys.path.insert(0, "$R")
import torch
from bench import make_volume
from pyradiomics_amd import engine
img, mask = make_volume($N, 32, "$dist", 0, torch.device("cuda", 0))
for i in range(2):
    engine.glszm_compact(img, mask, 32, $N ** 3)
torch.cuda.synchronize()
PY
pass() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o $name -- python /tmp/glszm_pmc_run.py > $out/$name.log 2>&1
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_FLAT
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $out glszm > $R/gpurun_out/pmc_glszm_$tag.md
rm -rf $out
cat $R/gpurun_out/pmc_glszm_$tag.md
