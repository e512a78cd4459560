#!/usr/bin/env python3
"""Prints the last kernel dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite) as a timeline: start, end, queue, name.
usage: tools/rocpd_timeline.py <results.db> [n_last]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t]
    view = "kernels" if "kernels" in tabs else None
    if view:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
        q = "select start, end, %s, name from kernels order by start" % ("queue_id" if "queue_id" in cols else "0")
        rows = list(db.execute(q))
    else:
        print("tables:", tabs)
        return
    rows = rows[-n:]
    t0 = rows[0][0]
    for s, e, qid, name in rows:
        print("%10.1f %10.1f %8.1f us  q%-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, qid, name[:70]))


if __name__ == "__main__":
    main()
