import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:70],r.get("Queue_Id","")) for r in rows]
ev.sort()
# last 200 ms of the trace
tend=max(e[1] for e in ev)
win=[e for e in ev if e[1]>tend-300e6]
from collections import defaultdict
d=defaultdict(lambda:[0,0,set()])
for s,e,n,q in win:
    d[n][0]+=1; d[n][1]+=e-s; d[n][2].add(q)
for n,(c,t,q) in sorted(d.items(), key=lambda kv:-kv[1][1])[:15]:
    print(f"{n:70s} {c:5d} {t/1e6:9.2f} ms queues {sorted(q)}")
long=[e for e in ev if e[1]-e[0]>50e6]
print("kernels longer than 50 ms:", [(n,(e-s)/1e6,q) for s,e,n,q in long][:10])
