#!/usr/bin/env python3
"""Per-kernel totals out of a rocprofv3 results database (rocprofv3 --kernel-trace writes <dir>/<host>/<pid>_results.db):
kernel name, launches, total and average duration.  `python scripts/kernel_table.py <dir> [steps]`."""
import glob
import sqlite3
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for db in glob.glob(d + "/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# %s: %.2f ms of kernel time in total (%.2f ms per step)" % (db, tot, tot / steps))
    print("name,calls,total_ms,ms_per_step,avg_us,percent")
    for r in rows:
        print('"%s",%d,%.3f,%.3f,%.1f,%.1f' % (r[0], r[1], r[2], r[2] / steps, r[3], 100 * r[2] / tot))
