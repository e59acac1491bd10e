#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd sqlite database, as CSV --
the same columns `rocprofv3 --stats --output-format csv` writes.  Usage: rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) '
                  'from kernels group by name order by 3 desc').fetchall()
tot = sum(r[2] for r in rows)
out = csv.writer(open(sys.argv[2], 'w', newline='') if len(sys.argv) > 2 else sys.stdout)
out.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
for n, c, t, a, mn, mx in rows:
    out.writerow([n, c, t, '%.1f' % a, '%.2f' % (100.0 * t / tot), mn, mx])
