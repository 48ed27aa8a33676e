#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (kernel trace +/- PMC) as plain text for profiles/.
usage: rocpd_summary.py <results.db> [title]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    print("# " + title)
    print("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite), durations in microseconds")
    rows = cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1.0
    print("%-8s %-12s %-12s %-10s %-10s %-6s %-6s %-6s %-7s %-8s %-9s %-5s %-6s %s" % (
        "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr", "lds_B", "scratch", "wg", "grid_x", "kernel"))
    for r in rows:
        print("%-8d %-12.1f %-12.3f %-10.3f %-10.3f %-6.2f %-6d %-6d %-7d %-8d %-9d %-5d %-6d %s" % (
            r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[12] or 0,
            r[11] or 0, r[0]))
    try:
        pmc = cur.execute("select name, counter_name, sum(value), count(*) from counters_collection group by name, counter_name "
                          "order by name, counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n# PMC counters (sum over dispatches, dispatch count)")
        for r in pmc:
            print("%-28s %-22.6g %-6d %s" % (r[1], r[2], r[3], r[0]))


if __name__ == "__main__":
    main()
