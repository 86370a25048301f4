#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`) as a markdown
table: per-kernel calls / total / average duration (us) and share.  Usage: rocpd_stats.py results.db [filter]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows:
        short = name.split("(")[0].replace("void ", "")
        if len(short) > 90:
            short = short[:87] + "..."
        if flt and flt not in name:
            continue
        print("| `%s` | %d | %.1f | %.2f | %.2f |" % (short, calls, total, avg, pct))


if __name__ == "__main__":
    main()
