"""Dump the per-kernel statistics of a rocprofv3 rocpd database (trace_results.db) as CSV."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
with open(out, 'w') as f:
    f.write('"Name","Calls","TotalDurationUs","AverageUs","Percentage"\n')
    for r in rows:
        f.write('"%s",%d,%.3f,%.3f,%.3f\n' % r)
print(open(out).read())
