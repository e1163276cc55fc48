import sys, collections, statistics as st
wg=[]
for l in open(sys.argv[1]):
    if l.startswith('#') or l.startswith('C'): continue
    v=list(map(int,l.split())); 
    if v[3]-v[2] > 200: wg.append(v)
cu=collections.defaultdict(list)
for w in wg:
    hw,xcc=w[6],w[7]&0xF
    cu[(xcc,(hw>>13)&7,(hw>>12)&1,(hw>>8)&0xF)].append(w)
# for each WG compute mean co-resident count over its life
rows=[]
for k,v in cu.items():
    for w in v:
        s,e=w[2],w[3]
        ov=0
        for o in v:
            ov+=max(0,min(e,o[3])-max(s,o[2]))
        rows.append((s/100,(e-s)/100,ov/(e-s),w[1]))
rows.sort()
# bucket by start time
import math
b=collections.defaultdict(list)
for s,d,c,m in rows: b[int(s//10)].append((d,c))
for k in sorted(b):
    ds=[x[0] for x in b[k]]; cs=[x[1] for x in b[k]]
    print(f"start {k*10:4d}-{k*10+10:4d} us: n={len(ds):4d} duration med {st.median(ds):6.1f} us, mean co-residency {sum(cs)/len(cs):.2f}, duration/co-res {st.median(ds)/(sum(cs)/len(cs)):.1f}")
