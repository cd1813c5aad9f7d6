"""Summaries of a rocprofv3 rocpd (SQLite) result file.
usage: python tools/rocpd_summary.py kernels  <results.db>   # --kernel-trace --stats style table
       python tools/rocpd_summary.py counters <results.db>   # --pmc totals per kernel and counter"""
import re, sqlite3, sys

mode, path = sys.argv[1], sys.argv[2]
c = sqlite3.connect(path)
short = lambda n: re.sub(r"\(.*", "", n)
if mode == "kernels":
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for n, k, tot, avg, lo, hi in rows:
        print('"%s",%d,%d,%.3f,%.2f,%d,%d' % (short(n), k, tot, avg, 100.0 * tot / total, lo, hi))
else:
    rows = c.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by 1, 2 order by 1, 2").fetchall()
    print("kernel,counter,launches,total,per_launch")
    for n, cn, k, tot in rows:
        print('"%s",%s,%d,%.3f,%.3f' % (short(n), cn, k, tot, tot / k))
