#!/bin/bash
# round 5, second sitting: rocprofv3 kernel stats of the bench command after the x-angle kernels changed and the two-table walk
# (64 levels) joined the deferred pipeline -- its launches now sit on ONE stream, so the 64-level averages below are kernel
# speeds (until round 5a they overlapped on two streams).  The PMC passes of scripts/prof_r05.sh (profiles/r05_pmc.md,
# r05_counters.json) describe sweep_fw_kernel, which did not change.  Output: gpurun_out/r05b/kernel_stats.md
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BA="--no-cpu-baseline --no-modes --no-host-boundary"
{
echo "# r05b -- rocprofv3 --kernel-trace --stats of bench.py (scripts/prof_r05b.sh), final source of round 5"
echo
echo "Pipeline mode (default): the launches of a step sit on one stream; '<..., true>' of sweep_fw_kernel / sweep_fw2_kernel is the launch"
echo "that walks the 12 line angles of volume N-1 AND packs volume N; '<..., false>' are the synchronous calls and the flush."
echo "(bench.py runs 80 untimed device warm-up steps first: the averages include the process's first ~40 launches, which are 7 % slower"
echo "than the steady state -- profiles/r05b_probes.md section 6; kernel_ms in the line under each heading is the HIP-event figure of"
echo "the 20 timed launches of the same profiled run)"
echo
} > $O/kernel_stats.md
for lv in 32 64; do
  for d in uniform smooth; do
    rocprofv3 --kernel-trace --stats -d $O/stats_${lv}_$d -o s -- python $R/bench.py --steps 20 --warmup 3 $BA --dist $d --levels $lv > $O/stats_${lv}_$d.log 2>&1
    { echo "## bench.py --steps 20 --warmup 3 --levels $lv --dist $d"; tail -1 $O/stats_${lv}_$d.log | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"rows_ms": [0-9.]*\|"finalize_ms": [0-9.]*' | tr '\n' ' '; echo; echo;
      python $R/scripts/rocpd_stats.py $O/stats_${lv}_$d/s_results.db | grep -E "prad|rocclr|kernel \||---"; echo;
      if [ $d = uniform ]; then
        echo "per-dispatch durations of the walk + pack launch in launch order (us): the first 8 of the process, then the last 33 = the 3 warm-up + 20 timed + 10 instrumented steps";
        echo; echo '```';
        python - <<PY
import sqlite3
db = sqlite3.connect("$O/stats_${lv}_$d/s_results.db")
rows = [r[0] / 1e3 for r in db.execute("select duration from kernels where name like '%sweep_fw%kernel%' and name like '%true>%' and name not like '%rows%' order by start")]
print("first 8:", " ".join("%.1f" % v for v in rows[:8]))
print("last 33:", " ".join("%.1f" % v for v in rows[-33:]))
t = rows[-30:-10]
print("the 20 timed launches: mean %.2f  min %.1f  max %.1f" % (sum(t) / len(t), min(t), max(t)))
PY
        echo '```'; echo;
      fi; } >> $O/kernel_stats.md
  done
done
cd $R
find $O -name "*.db" -delete
find $O -name "*.csv" -size +200k -delete
cat $O/kernel_stats.md
