#!/bin/bash
# HBM-side bytes of WHOLE training steps (tools/train_once.py: 3 steps at T=27, B=256) from two rocprofv3 --pmc passes; summary -> gpurun_out/pmc_step.json
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcs_f -- python3 $R/tools/train_once.py 27 256 > $O/pmcs_f.log 2>&1 || echo "fetch pass failed"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcs_w -- python3 $R/tools/train_once.py 27 256 > $O/pmcs_w.log 2>&1 || echo "write pass failed"
cd $R && python - <<'PY'
import collections, csv, glob, json, os, re
def load(d, counter):
    f = max(glob.glob(f"{d}/*/*counter_collection.csv"), key=os.path.getmtime)
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k = m.group(1) if m else "other (torch / runtime)"
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    return acc, n
fe, nf = load("gpurun_out/pmcs_f", "FETCH_SIZE"); wr, nw = load("gpurun_out/pmcs_w", "WRITE_SIZE")
steps = 3
ks = sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, 0) * 2048 + wr.get(k, 0) * 1024))
per = {k: {"launches_per_step": nf[k] / steps, "fetch_GB_per_step": fe.get(k, 0) * 2048 / steps / 1e9, "write_GB_per_step": wr.get(k, 0) * 1024 / steps / 1e9} for k in ks}
tot_f = sum(v["fetch_GB_per_step"] for v in per.values()); tot_w = sum(v["write_GB_per_step"] for v in per.values())
json.dump({"workload": "3 training steps, B=256, T=27, bf16, 26 layers (tools/train_once.py); includes model construction / first-step packing launches",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read tally)", "fetch_GB_per_step": tot_f, "write_GB_per_step": tot_w,
           "hbm_GB_per_step": tot_f + tot_w, "per_kernel": per}, open("gpurun_out/pmc_step.json", "w"), indent=1)
print("whole step: fetch %.1f GB + write %.1f GB = %.1f GB" % (tot_f, tot_w, tot_f + tot_w))
PY
