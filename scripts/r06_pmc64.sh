R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_pmc64; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-modes --no-host-boundary --levels 64"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python $R/bench.py --steps 4 --warmup 2 --device-warmup-ms 0 $BA > $O/pmc_$name.log 2>&1; }
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
python $R/scripts/pmc_summary.py $O/pmc sweep_fw2 > $O/pmc.md
rm -rf $O/pmc
cat $O/pmc.md
