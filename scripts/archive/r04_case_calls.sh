#!/bin/bash
# round 4: launches per 256^3 case by kernel name and queue (rocprofv3 kernel trace of scripts/case_latency.py, case pipeline)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cs
CASE_ONLY=1 CASE_REP=9 rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o s -- python $R/scripts/case_latency.py > /tmp/cs.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/cs/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "swt3_fused_kernel" in r["Kernel_Name"])
lo, hi = marks[2], marks[-1]
ncase = len(marks) - 3
c = collections.Counter(); t = collections.Counter()
for r in rows:
    s = int(r["Start_Timestamp"])
    if lo <= s < hi:
        k = (r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:70])
        c[k] += 1; t[k] += int(r["End_Timestamp"]) - s
print("launches per case (queue, kernel): count, us per case")
for k, n in sorted(c.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("  q%s %-70s %6.1f %8.1f" % (k[0], k[1], n / ncase, t[k] / 1e3 / ncase))
PY
