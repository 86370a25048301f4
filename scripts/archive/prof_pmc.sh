#!/bin/bash
# PMC counter passes for the sweep kernels (separate passes; never combined with --sys-trace etc.)
# usage: prof_pmc.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
pass() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o $name -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "${BENCH_ARGS[@]}" > $out/$name.log 2>&1
}
BENCH_ARGS=("$@")
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
ls -R $out | head -40
