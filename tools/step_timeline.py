"""Kernel timeline (start, duration, queue) of one graph-replayed step from a rocprofv3 kernel trace."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_adam' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
span = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # optimiser launches per step (UDAGCN: 2 rounds)
a, b = idx[k], idx[k + span]
t0 = int(rows[a]['End_Timestamp']); qs = {}
for r in rows[a + 1:b + 1]:
    q = r['Queue_Id']; qs.setdefault(q, len(qs))
    s = (int(r['Start_Timestamp']) - t0) / 1000; e = (int(r['End_Timestamp']) - t0) / 1000
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    print(f"{s:8.1f} {e - s:6.1f} q{qs[q]} {name}")
