"""Mean per launch of the SQ counters collected by tools/pmc_sq.sh for the two MLP kernels -> JSON on stdout."""
import collections, csv, glob, json, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{d}/pmc[0-9]/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for key in ("k_mlp_fwd_s", "k_mlp_bwd_s", "k_mlp_fwd_r", "k_mlp_bwd_q"):
            if key in r["Kernel_Name"]:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}, indent=1))
