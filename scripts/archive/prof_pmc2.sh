#!/bin/bash
# usage: prof_pmc2.sh <variant> ; PMC passes on an ablation build
n=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc2_$n
mkdir -p $out
pass() { name=$1; shift
  PRAD_LIB=$R/build_variants/lib_$n.so PRAD_BENCH_NOCHECK=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o $name -- \
     python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
pass c SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM
pass d GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $out sweep_lines
