"""Summarise a rocprofv3 --pmc output directory: per kernel, mean counter value per dispatch."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: [0.0, 0])
for f in files:
    for row in csv.DictReader(open(f)):
        k = (row.get('Kernel_Name', '?')[:60], row.get('Counter_Name', '?'))
        agg[k][0] += float(row.get('Counter_Value', 0) or 0)
        agg[k][1] += 1
for (kern, ctr), (tot, n) in sorted(agg.items()):
    if 'gnnpp' in kern:
        print('%-62s %-34s mean/dispatch=%.4g  dispatches=%d' % (kern, ctr, tot / n, n))
