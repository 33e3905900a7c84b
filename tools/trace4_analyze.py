import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-24:],int(r['Grid_Size_X'])) for r in rows]
ev.sort()
k1=[e for e in ev if 'k1_demod' in e[2]]
n=len(k1)
# timed region = K1 launches [4, n-4): skip warm-up (4) and calibration (4)
t0=k1[8][0]; t1=k1[n-8][0]
win=[e for e in ev if e[0]>=t0 and e[1]<=t1]
steps=(n-16)/4
def union(iv):
    cur=None; b=0
    for s,e in sorted(iv):
        if cur is None: cur=[s,e]
        elif s<=cur[1]: cur[1]=max(cur[1],e)
        else: b+=cur[1]-cur[0]; cur=[s,e]
    if cur: b+=cur[1]-cur[0]
    return b
print(f"window {(t1-t0)/1e6:.1f} ms = {steps:.0f} steps -> {(t1-t0)/1e6/steps:.1f} ms/step under rocprof")
print(f"  any kernel resident      : {union([(s,e) for s,e,_,_ in win])/1e6/steps:6.1f} ms/step")
print(f"  K1 resident              : {union([(s,e) for s,e,n_,_ in win if 'k1_demod' in n_])/1e6/steps:6.1f} ms/step")
print(f"  big framer kernel resident: {union([(s,e) for s,e,n_,g in win if 'k2_' in n_ and g>=30000])/1e6/steps:6.1f} ms/step")
print(f"  neither K1 nor big framer : {((t1-t0)-union([(s,e) for s,e,n_,g in win if 'k1_demod' in n_ or ('k2_' in n_ and g>=30000)]))/1e6/steps:6.1f} ms/step")
tot=collections.Counter()
for s,e,n_,g in win: tot[n_ + (' (rerun)' if ('k2_' in n_ and g<30000) else '')]+=e-s
for n_,v in tot.most_common(8): print(f"  {n_:34s} sum {v/1e6/steps:8.2f} ms/step")
