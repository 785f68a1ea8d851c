#!/usr/bin/env python3
"""Summarise gpurun_out/sq/*.csv: per kernel, average of each SQ counter over its launches."""
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/sq/*.csv')):
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']
        if 'seamd::' not in nm or 'ntt_polys' in nm: continue
        key = nm.split('(')[0].replace('void ', '').replace('seamd::', '').strip()
        acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-28s %16.0f" % (c, sum(v) / len(v)))
