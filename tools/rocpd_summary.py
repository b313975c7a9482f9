#!/usr/bin/env python3
"""Developer tool: turn rocprofv3's rocpd (sqlite) output into the small
summaries kept under profiles/.

  rocpd_summary.py stats  <results.db> <out.csv>       kernel-trace --stats table
  rocpd_summary.py pmc    <results.db> <kernel-substr>  per-dispatch counter sums (JSON on stdout)
"""
import csv
import json
import sqlite3
import sys


def stats(db, out):
    con = sqlite3.connect(db)
    cur = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in cur.fetchall():
            w.writerow(r)


def pmc(db, substr):
    con = sqlite3.connect(db)
    cur = con.execute(
        "select dispatch_id, counter_name, sum(value), max(duration) from counters_collection "
        "where kernel_name like ? group by dispatch_id, counter_name order by dispatch_id",
        ("%" + substr + "%",))
    rows = [{"dispatch_id": r[0], "counter": r[1], "value": r[2], "duration_ns": r[3]} for r in cur.fetchall()]
    json.dump(rows, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3])
