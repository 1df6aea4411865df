import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'k_cg_resident' in r['Kernel_Name']]
tot=[]
for a,b in zip(idx[2:-1],idx[3:]):
    tot.append((int(rows[b]['End_Timestamp'])-int(rows[a]['End_Timestamp']))/1000)
print("passes", [round(t,1) for t in tot])
a,b=idx[-2],idx[-1]
prev_end=None
for r in rows[a:b+1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1000 if prev_end else 0
    print(f"{r['Kernel_Name'][:60]:60s} dur {(e-s)/1000:8.1f} us  gap {gap:6.1f}")
    prev_end=e
