#!/usr/bin/env python3
"""Dumps the per-kernel statistics of a rocprofv3 run (rocpd sqlite output, `--kernel-trace --stats`) as text.
usage: tools/rocpd_summary.py <results.db> [<out.txt>]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["%-12s %8s %14s %12s  %s" % ("pct", "calls", "total_us", "avg_us", "kernel")]
    for name, calls, total, avg, pct in rows:
        lines.append("%-12.2f %8d %14.1f %12.1f  %s" % (pct, calls, total, avg, name))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
