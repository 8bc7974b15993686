import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*kernel_trace.csv'))[-1]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-int(sys.argv[2]):] if len(sys.argv)>2 else rows[-31:]
t0=int(rows[0]['Start_Timestamp']); qs={}
for r in rows:
    q=r['Queue_Id']; qs.setdefault(q,len(qs))
    s=(int(r['Start_Timestamp'])-t0)/1000; e=(int(r['End_Timestamp'])-t0)/1000
    print(f"{s:8.1f} {e-s:6.1f} q{qs[q]} grid={r['Grid_Size_X']:>6} {r['Kernel_Name'][:50]}")
