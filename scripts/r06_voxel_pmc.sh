#!/bin/bash
# round 6: SQ counters of the sliding-window map kernel (256^3, 5^3 and 5 x 5 windows, JointEntropy); two passes, kernel trace only.
# usage (on the GPU box): scripts/r06_voxel_pmc.sh  -> gpurun_out/r06_voxpmc/pmc.md
R=$(cd $(dirname $0)/.. && pwd); O=$R/gpurun_out/r06_voxpmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python $R/scripts/r06_voxel3d.py 256 > $O/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $O/pmc voxel_glcm_slide > $O/pmc.md
cat $O/pmc.md
