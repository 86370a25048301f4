cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/case_prof -o c -- python $R/scripts/batch_toggle.py > $R/gpurun_out/case_prof.log 2>&1
tail -2 $R/gpurun_out/case_prof.log
python $R/scripts/rocpd_stats.py $R/gpurun_out/case_prof/c_results.db | head -40
find $R/gpurun_out/case_prof -name "*.db" -delete
