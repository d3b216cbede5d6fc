"""Per-layer timeline of a three-stream training step from a rocprofv3 kernel trace: start - end (us, relative to the layer's first kernel) of every launch by
hardware queue, for one backward and one forward layer, and the mean layer durations.   python tools/layer_timeline.py <..._kernel_trace.csv>
(bash tools/prof27_3s.sh writes gpurun_out/prof27_3s/t_kernel_trace.csv)"""
import csv, re, collections, sys
path=sys.argv[1]
rows=[(r["Kernel_Name"],int(r["Start_Timestamp"]),int(r["End_Timestamp"]),int(r["Queue_Id"])) for r in csv.DictReader(open(path))]
rows.sort(key=lambda r:r[1])
starts=[i for i,r in enumerate(rows) if "k_prologue_fwd" in r[0]]; ends=[i for i,r in enumerate(rows) if "k_adamw" in r[0]]
step=rows[starts[-1]:ends[-1]+1]
def short(n):
    m=re.search(r"(k_[a-z0-9_]+)",n); return m.group(1) if m else n[:30]
gb=[i for i,r in enumerate(step) if 'k_gate_bwd' in r[0]]
durs=[]
for li in range(len(gb)-1):
    seg=step[gb[li]:gb[li+1]]; durs.append((seg[-1][2]-seg[0][1])/1e3)
print('backward layer durations us: mean',sum(durs)/len(durs))
for li in (5,):
    seg=step[gb[li]:gb[li+1]]
    byq=collections.defaultdict(list)
    for r in seg: byq[r[3]].append(r)
    for q,rs in sorted(byq.items()):
        print(' queue',q,' '.join(f"{short(r[0])[2:12]}:{(r[1]-seg[0][1])/1e3:.0f}-{(r[2]-seg[0][1])/1e3:.0f}" for r in rs))
gf=[i for i,r in enumerate(step) if 'k_gate_fwd' in r[0]]
d2=[(step[gf[i+1]][2]-step[gf[i]][2])/1e3 for i in range(len(gf)-1)]
print('forward layer durations us: mean',sum(d2)/len(d2))
seg=step[gf[5]+1:gf[6]+1]
byq=collections.defaultdict(list)
for r in seg: byq[r[3]].append(r)
for q,rs in sorted(byq.items()):
    print(' queue',q,' '.join(f"{short(r[0])[2:12]}:{(r[1]-seg[0][1])/1e3:.0f}-{(r[2]-seg[0][1])/1e3:.0f}" for r in rs))
