import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for i, r in enumerate(rows):
    name = r['Kernel_Name'][:50]
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    gap = int(r['Start_Timestamp']) - int(rows[i-1]['End_Timestamp']) if i else 0
    by[(name, r['Grid_Size_X'])].append((d, gap, i))
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 5: continue
    v2 = v[len(v)//2:]
    ds = sorted(x[0] for x in v2); gs = sorted(x[1] for x in v2)
    print(f"{len(v):6d} x {k[0]:50s} grid {k[1]:>8s}: dur median {ds[len(ds)//2]/1e3:7.2f} us  p10 {ds[len(ds)//10]/1e3:7.2f}  p90 {ds[9*len(ds)//10]/1e3:7.2f}   gap before: median {gs[len(gs)//2]/1e3:6.2f} us")
n = len(rows)
seg = rows[n - 1200:n - 1178]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.2f} +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.2f}  {r['Kernel_Name'][:34]} grid {r['Grid_Size_X']}")
