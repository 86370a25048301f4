#!/usr/bin/env python
"""Per-dispatch durations of one kernel from a rocprofv3 rocpd database: rocpd_dispatches.py results.db <substring> [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = db.execute("select name, start, duration, grid_x, workgroup_x, vgpr_count, lds_size from kernels "
                  "where name like ? order by start", ("%" + pat + "%",)).fetchall()
print("%d dispatches of *%s*" % (len(rows), pat))
for name, start, dur, gx, wx, vg, lds in rows[-n:]:
    print("%-40s %9.1f us  grid %d x %d  vgpr %d lds %d" % (name.split("(")[0][-40:], dur / 1e3, gx // max(wx, 1), wx, vg, lds))
