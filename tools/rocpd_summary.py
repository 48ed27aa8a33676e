#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (kernel trace +/- PMC) as plain text for profiles/.
usage: rocpd_summary.py <results.db> [title] [--skip-first N]

--skip-first N (round 5): the profiled command ran N warm-up steps before its timed region; the first N launches of every
kernel that is launched more than N times are listed but kept out of the `timed_avg_us` / `median_us` columns, so that the
summary reproduces the bench line's avg_launch_ms (which excludes the warm-up) instead of averaging cold launches in."""
import sqlite3
import statistics
import sys


def main():
    argv = list(sys.argv[1:])
    skip = 0
    if "--skip-first" in argv:
        i = argv.index("--skip-first")
        skip = int(argv[i + 1])
        del argv[i:i + 2]
    db = sqlite3.connect(argv[0])
    cur = db.cursor()
    title = argv[1] if len(argv) > 1 else argv[0]
    print("# " + title)
    print("# source: rocprofv3 --kernel-trace --stats (rocpd sqlite), durations in microseconds")
    timed = {}
    for name, dur in cur.execute("select name, duration/1e3 from kernels order by start").fetchall():
        timed.setdefault(name, []).append(dur)
    print("# timed_avg_us / median_us: without the first %d launches of a kernel (the command's warm-up steps)" % skip)
    print("%-8s %-12s %-12s %s" % ("timed", "timed_avg_us", "median_us", "kernel"))
    for name, d in sorted(timed.items(), key=lambda kv: -sum(kv[1])):
        t = d[skip:] if len(d) > skip else d
        print("%-8d %-12.3f %-12.3f %s" % (len(t), sum(t) / len(t), statistics.median(t), name))
    print("# every launch:")
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
