import sys, collections
rows=[]
for l in open(sys.argv[1]):
    p=l.split(None,4)
    rows.append((int(p[0]),int(p[1]),p[2],p[3],p[4].strip()))
# take the last 40% window (steady state under load)
tmax=max(r[1] for r in rows)
big=[r for r in rows if ('conv_group' in r[4] or 'pair_group' in r[4] or 'rb_group' in r[4])]
print("big kernels", len(big))
# find steady-state window: last region where queue ids vary
qs=collections.Counter(r[2] for r in rows); print("queues", qs)
st=collections.Counter(r[3] for r in rows); print("streams", len(st))
# window = last 60 ms — or, with a kernel-name substring as the second argument, the 60 ms that end with the last launch of such a
# kernel (bench.py ends on vocoder-only legs: "gate16" / "wn_layer" selects the last region of FULL calls)
if len(sys.argv) > 2:
    tmax=max(r[1] for r in rows if any(k in r[4] for k in sys.argv[2].split(",")))
w0=tmax-60_000_000 if tmax>80_000_000 else tmax//2
sel=[r for r in rows if r[0]>=w0 and r[1]<=tmax]
bigs=[r for r in sel if ('conv_group' in r[4] or 'pair_group' in r[4] or 'rb_group' in r[4])]
print("window kernels", len(sel), "big", len(bigs), "window ms", (tmax-w0)/1e6)
# per-kernel-name average duration in window
d=collections.defaultdict(list)
for r in sel: d[r[4]].append(r[1]-r[0])
tot=0
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:14]:
    print(f"{n[:70]:70s} n={len(v):5d} avg {sum(v)/len(v)/1e3:8.1f} us  total {sum(v)/1e6:7.2f} ms")
    tot+=sum(v)
print("sum of all kernel durations in window (ms):", sum(sum(v) for v in d.values())/1e6)
# concurrency of big kernels over time
ev=[]
for r in bigs: ev+= [(r[0],1),(r[1],-1)]
ev.sort()
cur=0; last=ev[0][0]; hist=collections.Counter()
for t,dl in ev:
    hist[cur]+=t-last; last=t; cur+=dl
T=sum(hist.values())
print("big-kernel concurrency histogram (fraction of time):", {k: round(v/T,3) for k,v in sorted(hist.items())})
# all kernels
ev=[]
for r in sel: ev+= [(r[0],1),(r[1],-1)]
ev.sort()
cur=0; last=ev[0][0]; hist=collections.Counter()
for t,dl in ev:
    hist[cur]+=t-last; last=t; cur+=dl
T=sum(hist.values())
print("all-kernel concurrency histogram:", {k: round(v/T,3) for k,v in sorted(hist.items())})
