"""Summarise rocprofv3 --pmc counter_collection CSVs under a directory: mean of every counter over
the dispatches of each kernel (name shortened).  usage: pmc_summary.py DIR [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else 'conv_apply'
for f in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get('Kernel_Name', '')
            if want not in k:
                continue
            acc[k[:60]][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', os.path.relpath(f, root))
    for k, cs in acc.items():
        for c, v in sorted(cs.items()):
            print('  %-62s %-36s n=%-4d mean=%.6g' % (k, c, len(v), sum(v) / len(v)))
