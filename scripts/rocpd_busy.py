#!/usr/bin/env python
"""GPU busy time from a rocprofv3 rocpd database: union of the kernel intervals vs their sum vs the span from the first
start to the last end (how well concurrent streams fill the GPU).  rocpd_busy.py results.db [skip_first_fraction]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, start + duration, name from kernels order by start").fetchall()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + skip * (t1 - t0)
rows = [r for r in rows if r[0] >= cut]
span = max(r[1] for r in rows) - rows[0][0]
total = sum(b - a for a, b, _ in rows)
busy, cur_a, cur_b = 0, None, None
for a, b, _ in rows:
    if cur_b is None or a > cur_b:
        if cur_b is not None:
            busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
print("kernels %d  span %.2f ms  union-busy %.2f ms (%.1f %% of span)  sum of durations %.2f ms (%.2fx the span)"
      % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, total / 1e6, total / span))
agg = {}
for a, b, n in rows:
    k = n.split("(")[0].replace("void ", "")[:70]
    agg[k] = agg.get(k, 0) + (b - a)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:28]:
    print("  %-72s %8.2f ms  %5.1f %%" % (k, v / 1e6, 100.0 * v / total))
