#!/bin/bash
# usage: prof_case_kernels.sh <tag> -- rocprofv3 kernel trace of whole cases (scripts/prof_case_engine.py): slowest dispatches
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/case_$tag -o $tag -- python $R/scripts/prof_case_engine.py > $R/gpurun_out/case_$tag.log 2>&1
tail -10 $R/gpurun_out/case_$tag.log
python - <<PY
import sqlite3
db = sqlite3.connect("$R/gpurun_out/case_$tag/${tag}_results.db")
rows = db.execute("select name, duration, start from kernels order by duration desc limit 14").fetchall()
for n, d, s in rows:
    print("%-70s %9.1f us" % (n.split("(")[0][-70:], d / 1e3))
PY
python $R/scripts/rocpd_stats.py $R/gpurun_out/case_$tag/${tag}_results.db | head -16
