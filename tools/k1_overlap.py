#!/usr/bin/env python3
"""What slows a k1_demod2 launch down?  Least-squares fit of every K1 launch's duration in a rocprofv3
kernel trace on the fraction of its time each other kernel type was resident beside it.
usage: tools/k1_overlap.py [kernel_trace.csv]   (default: profiles/r01_bench_kernel_trace.csv)"""
import sys, os
import csv, collections, numpy as np
rows=[]
for r in csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r01_bench_kernel_trace.csv'))):
    n=r['Kernel_Name']
    for key in ('k1_demod2','k2_clock_rla','k2_clock','k2_rla','k3_bursts','k3_scan'):
        if key in n: rows.append((key,int(r['Start_Timestamp']),int(r['End_Timestamp']),int(r['Grid_Size_X']))); break
rows.sort(key=lambda x:x[1])
k1=[r for r in rows if r[0]=='k1_demod2']; others=[r for r in rows if r[0]!='k1_demod2']
print(collections.Counter((r[0],r[3]) for r in others).most_common(10))
kinds=['k2_clock','k2_clock_rla_main','k2_clock_rla_rerun','k3_bursts','k3_scan']
def kind(r):
    if r[0]=='k2_clock_rla': return 'k2_clock_rla_main' if r[3]>=60000 else 'k2_clock_rla_rerun'
    return r[0]
X=[];Y=[]
for a in k1:
    d=a[2]-a[1]; ov=collections.Counter()
    for b in others:
        if b[2]<=a[1] or b[1]>=a[2]: continue
        ov[kind(b)]+=(min(a[2],b[2])-max(a[1],b[1]))/d
    X.append([ov[k] for k in kinds]); Y.append(d/1e6)
X=np.array(X);Y=np.array(Y)
A=np.hstack([np.ones((len(Y),1)),X]); coef=np.linalg.lstsq(A,Y,rcond=None)[0]
print('K1 n=%d median %.2f; intercept %.2f'%(len(Y),np.median(Y),coef[0]))
for k,c,m in zip(kinds,coef[1:],X.mean(0)): print('%-20s coef %.2f  mean overlap %.2f -> %.2f ms'%(k,c,m,c*m))
pred=A@coef; print('R2 %.3f'%(1-((Y-pred)**2).sum()/((Y-Y.mean())**2).sum()))
# durations of other kernels
for k in kinds:
    d=[(r[2]-r[1])/1e6 for r in others if kind(r)==k]
    print(k,'n',len(d),'mean %.2f'%np.mean(d))
