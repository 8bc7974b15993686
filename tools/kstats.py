#!/usr/bin/env python
"""Print the top kernels of a rocprofv3 --stats output directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for r in list(csv.DictReader(open(f)))[:40]:
    if pat in r['Name']:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f}")
