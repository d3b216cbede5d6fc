"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/mlp_bench.py into profiles/<name>.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -- python3 $R/tools/mlp_bench.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- python3 $R/tools/mlp_bench.py
    python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/r1_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores and float atomics.
"""
import collections, csv, glob, json, os, re, sys


def per_kernel(d, counter):
    f = max(glob.glob(f"{d}/*/*counter_collection.csv"), key=os.path.getmtime)     # newest run in the directory
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
            if m:
                acc[m.group(0)].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = per_kernel(fdir, "FETCH_SIZE"), per_kernel(wdir, "WRITE_SIZE")
    res = {"unit": "bytes per launch", "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read tally)", "workload": "M = 256*27*17 = 117,504 tokens (bench.py config)",
           "kernels": {}}
    for k in sorted(fe):
        res["kernels"][k] = {"launches_sampled": fe[k][1], "fetch_bytes": fe[k][0] * 1024 * 2, "write_bytes": wr.get(k, (0, 0))[0] * 1024}
        res["kernels"][k]["hbm_bytes"] = res["kernels"][k]["fetch_bytes"] + res["kernels"][k]["write_bytes"]
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
