import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-22:],int(r['Grid_Size_X']),r.get('Stream_Id','?'),r.get('Queue_Id','?')) for r in rows]
ev.sort()
k1=[e for e in ev if 'k1_demod' in e[2] and e[3]>1000]
# pick one queue/stream and print its timeline between its 3rd and 4th big K1 launches
qs=collections.Counter(e[5] for e in k1)
q=qs.most_common()[0][0]
mine=[e for e in ev if e[5]==q]
kk=[e for e in mine if 'k1_demod' in e[2] and e[3]>1000]
t0=kk[3][0]; t1=kk[4][0]
print(f"queue {q}: one push = {(t1-t0)/1e6:.2f} ms (K1 start to next K1 start)")
prev=t0
for s,e,n,g,st,qq in mine:
    if s<t0 or s>=t1: continue
    if e-s>50000 or s-prev>300000:
        print(f"  +{(s-t0)/1e6:7.2f}  gap {(s-prev)/1e6:6.2f}  dur {(e-s)/1e6:6.2f}  {n:22s} grid {g}")
    prev=max(prev,e)
