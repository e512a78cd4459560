#!/usr/bin/env python3
"""Per-kernel averages of one PMC counter from rocprofv3's rocpd database (separate --pmc passes, tools/profile_round.sh).
usage: tools/rocpd_pmc_summary.py <results.db> [<results.db> ...]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(\w+(?:<\w+>)?)\(", name)
    return m.group(1) if m else name[:40]


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
                      "group by kernel_name, counter_name, dispatch_id").fetchall()
    agg = {}
    for k, c, _, v in rows:
        agg.setdefault((short(k), c), []).append(v)
    for (k, c), vs in sorted(agg.items(), key=lambda t: -sum(t[1])):
        print("%-28s %-10s  calls=%3d avg=%12.1f KB min=%12.1f max=%12.1f" % (k, c, len(vs), sum(vs) / len(vs), min(vs), max(vs)))
