#!/bin/bash
# round 4: PMC counters of the 64-level path (sweep_fw2_kernel, sweep_rows_kernel, pack_levels16_kernel), 512^3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_ng64
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-modes --no-host-boundary --levels 64"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python $R/bench.py --steps 3 --warmup 1 $BA > $O/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $O/pmc > $O/pmc.md
find $O -name "*.csv" -delete
cat $O/pmc.md
