import sys, statistics as st, collections
path=sys.argv[1]; every=61
K={0:11,1:7,2:3}
mfma_per_tap=int(sys.argv[2]) if len(sys.argv)>2 else 16   # MFMAs per wave per chunk per tap
wg={}
ch=collections.defaultdict(list)
for l in open(path):
    if l.startswith('#'): continue
    p=l.split()
    if p[0]=='C':
        s,w,c,s0,s1,s2,s3=map(int,p[1:])
        ch[(s,w)].append((c,s0,s1,s2,s3))
    else:
        v=list(map(int,p)); wg[v[0]]=v
res=collections.defaultdict(lambda: collections.defaultdict(list))
for (s,w),v in ch.items():
    lin=s*every+3
    if lin not in wg: continue
    m=wg[lin][1]
    v.sort()
    for i,(c,s0,s1,s2,s3) in enumerate(v):
        if s0 and s1: res[m]['mfma'].append(s1-s0)
        if s1 and s2: res[m]['store'].append(s2-s1)
        if s2 and s3: res[m]['bar'].append(s3-s2)
        if i+1<len(v) and v[i+1][1] and s0: res[m]['cycle'].append(v[i+1][1]-s0)
    # whole tile: first s0 to last s3
    res[m]['tile'].append(v[-1][4]-v[0][1]); res[m]['nch'].append(len(v))
for m in sorted(res):
    own=K[m]*mfma_per_tap*64
    r=res[m]
    med=lambda x: st.median(x) if x else 0
    print(f"member k={K[m]}: own MFMA issue per chunk {own} cyc; MFMA-loop phase med {med(r['mfma']):.0f} (x{med(r['mfma'])/own:.2f}); store {med(r['store']):.0f}; barrier {med(r['bar']):.0f}; chunk cycle med {med(r['cycle']):.0f} (x{med(r['cycle'])/own:.2f}); chunks/tile {med(r['nch'])}; n={len(r['mfma'])}")
print("--- means")
for m in sorted(res):
    own=K[m]*mfma_per_tap*64
    r=res[m]
    mean=lambda x: sum(x)/len(x) if x else 0
    gap=[]
for (s,w),v in ch.items():
    lin=s*every+3
    if lin not in wg: continue
    m=wg[lin][1]
    for a,b in zip(v,v[1:]):
        if a[4] and b[1]: res[m]['gap'].append(b[1]-a[4])
for m in sorted(res):
    own=K[m]*mfma_per_tap*64
    r=res[m]
    mean=lambda x: sum(x)/len(x) if x else 0
    print(f"k={K[m]}: own {own}; mean mfma-phase {mean(r['mfma']):.0f} store {mean(r['store']):.0f} barrier {mean(r['bar']):.0f} gap(after barrier -> next top) {mean(r['gap']):.0f}; mean cycle {mean(r['cycle']):.0f} = x{mean(r['cycle'])/own:.2f}")
