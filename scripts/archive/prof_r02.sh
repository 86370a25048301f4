#!/bin/bash
# round-2 evidence pack (final state): kernel stats with the default two lanes and with PRAD_LANES=1 (every launch alone
# on the GPU: the durations bench.py's roofline block uses), PMC passes (one lane), ablation table, one-angle-alone times
# + fabric FETCH per angle, 1-rank torch.distributed launch.  Everything lands in gpurun_out/r02b/ (copy the .md files to
# profiles/).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-modes --no-host-boundary"
rm -f $O/kernel_stats.md
for d in uniform smooth; do
 for L in 2 1; do
  PRAD_LANES=$L rocprofv3 --kernel-trace --stats -d $O/stats_${d}_$L -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --dist $d > $O/stats_${d}_$L.log 2>&1
  { echo "## PRAD_LANES=$L bench.py --steps 20 --warmup 3 --dist $d (rocprofv3 --kernel-trace --stats)"; tail -1 $O/stats_${d}_$L.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"overlapped_kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*\|"pack_ms": [0-9.]*' | tr '\n' ' '; echo; echo;
    python $R/scripts/rocpd_stats.py $O/stats_${d}_$L/s_results.db | grep -E "prad|rocclr|kernel \||---"; echo; } >> $O/kernel_stats.md
 done
done
rocprofv3 --kernel-trace --stats -d $O/stats_256 -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --size 256 > $O/stats_256.log 2>&1
{ echo "## bench.py --size 256"; tail -1 $O/stats_256.log | grep -o '"value": [0-9.]*\|"kernel_ms": [0-9.]*\|"pipeline_ms": [0-9.]*' | tr '\n' ' '; echo; echo; python $R/scripts/rocpd_stats.py $O/stats_256/s_results.db | grep -E "prad|kernel \||---"; } >> $O/kernel_stats.md
# PMC (separate passes, kernel-trace only)
pass() { name=$1; shift; PRAD_LANES=1 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc/$name -o $name -- python $R/bench.py --steps 2 --warmup 1 $BA > $O/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE
python $R/scripts/pmc_summary.py $O/pmc > $O/pmc.md
# ablations (sweep family time of the bench = sweep_fw_kernel; wrong results by design)
cd $R
{ echo "| build | uniform sweep ms | smooth sweep ms |"; echo "|---|---:|---:|";
for v in base nobump noload noasm noz; do
  L=""; [ $v != base ] && L="PRAD_LIB=$R/build_variants/lib_$v.so"
  u=$(env $L PRAD_BENCH_NOCHECK=1 python bench.py --steps 10 --warmup 3 $BA 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
  s=$(env $L PRAD_BENCH_NOCHECK=1 python bench.py --steps 10 --warmup 3 $BA --dist smooth 2>&1 | tail -1 | grep -o '"kernel_ms": [0-9.]*' | cut -d' ' -f2)
  echo "| $v | $u | $s |"
done; } > $O/ablation.md
python scripts/per_angle.py 512 > $O/per_angle.log 2>&1
python scripts/per_angle.py 512 --dist=smooth >> $O/per_angle.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_angle -o a -- python $R/scripts/per_angle.py 512 > $O/pmc_angle.log 2>&1
python - <<PY > $O/per_angle_fetch.md
import csv, glob
rows = []
for f in glob.glob("$O/pmc_angle/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep_fw_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r.get("Grid_Size", "")))
rows.sort()
print("| dispatch | grid | FETCH_SIZE KiB |"); print("|---:|---:|---:|")
for d, v, g in rows: print("| %d | %s | %.0f |" % (d, g, v))
PY
cd $R
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/torchrun_1rank.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_plain.log 2>&1
tail -1 $O/torchrun_1rank.log | cut -c1-400; tail -1 $O/bench_plain.log | cut -c1-300
cat $O/ablation.md; cat $O/per_angle.log | cut -c1-600
# only the summaries travel back (gpurun_out is capped at 64 MiB)
find $O -name "*.db" -delete
find $O -name "*.csv" -size +200k -delete
du -sh $O
