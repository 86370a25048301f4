#!/bin/bash
# PMC counters of the voxel-based GLCM map kernel (scripts/bench_voxel.py, 256^3, 5x5 window, JointEntropy)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/voxpmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -o a -- python $R/scripts/bench_voxel.py > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/b -o b -- python $R/scripts/bench_voxel.py > $O/b.log 2>&1
python $R/scripts/pmc_summary.py $O voxel_glcm
find $O -name "*.csv" -size +200k -delete
