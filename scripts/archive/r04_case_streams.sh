#!/bin/bash
# round 4: which stream bounds a 256^3 case?  rocprofv3 kernel trace of scripts/case_latency.py (case pipeline only), per-queue busy time
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cs
CASE_ONLY=1 CASE_REP=9 rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -o s -- python $R/scripts/case_latency.py > /tmp/cs.log 2>&1
grep enqueueSegment= /tmp/cs.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/cs/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "kernel records; columns:", list(rows[0].keys())[:14])
qk = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
t1 = max(int(r["End_Timestamp"]) for r in rows)
# steady state: from the third case on (every case launches the fused wavelet kernel exactly once)
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "swt3_fused_kernel" in r["Kernel_Name"])
lo = marks[2]
t1 = marks[-1]
rows = [r for r in rows if int(r["Start_Timestamp"]) < t1]
print("cases in the window:", len(marks) - 3)
per = collections.defaultdict(float); cnt = collections.Counter(); names = collections.defaultdict(collections.Counter)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < lo: continue
    q = r.get(qk, "?")
    per[q] += (e - s); cnt[q] += 1
    names[q][r["Kernel_Name"].split("(")[0][:60]] += (e - s)
span = t1 - lo
ncase = len(marks) - 3
print("steady-state span %.2f ms = %.2f ms per case (under the profiler); %d kernel + copy + fill records per case" % (span / 1e6, span / 1e6 / ncase, sum(cnt.values()) // ncase))
for q in sorted(per, key=lambda k: -per[k]):
    print("queue %s: %d records per case, busy %.2f ms per case = %.0f %% of the span" % (q, cnt[q] // ncase, per[q] / 1e6 / ncase, 100 * per[q] / span))
    for n, t in names[q].most_common(6):
        print("      %-60s %.2f ms per case" % (n, t / 1e6 / ncase))
# union of all kernel intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if int(r["Start_Timestamp"]) >= lo)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("GPU busy (union of kernels) %.0f %% of the span" % (100 * busy / span))
PY
