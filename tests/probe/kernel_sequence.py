import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'encoder_fwd_kernel' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s=int(r['Start_Timestamp'])-t0; d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    print("%8.1f %7.1f  %s  grid=%s" % (s/1e3, d/1e3, r['Kernel_Name'][:100], r.get('Grid_Size_X','')))
