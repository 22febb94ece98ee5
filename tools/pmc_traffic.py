"""Parse two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs with --kernel-trace only, as
MI355X_MICROARCH.md prescribes) of `bench.py --steps 4 --warmup 2` into profiles/<round>_traffic.json:
HBM bytes per launch of the two dominant kernels = FETCH_SIZE[KB] * 1024 * 2 (gfx950 tallies a 128-B read request
as 64 B) + WRITE_SIZE[KB] * 1024, with k_prep / k_adam as calibration against known byte counts.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys
from collections import defaultdict


def mean_by_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(list)
    for path in f:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = mean_by_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = mean_by_kernel(sys.argv[2], "WRITE_SIZE")
pick = lambda d, pat: next((v for k, v in d.items() if all(p in k for p in pat)), None)
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on bench.py --steps 4 --warmup 2 "
               "--no-cpu-baseline; bytes per launch = FETCH_SIZE[KB]*1024*2 (gfx950 counts a 128-B read request as 64 B: "
               "MI355X_MICROARCH.md, HBM) + WRITE_SIZE[KB]*1024"}
# the TN kernel name is shared by the grouped dW1 launch and the dE GEMM: take the launches with the larger fetch
def tn_split(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    vals = defaultdict(list)
    for path in f:
        for r in csv.DictReader(open(path)):
            if "seg_gemm_kernel" in r["Kernel_Name"] and "false, false" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return vals
for name, pat in (("MAIN", ("k_main_fwd",)), ("k_prep", ("k_prep",)), ("k_adam", ("k_adam",))):
    fk, wk = pick(fetch, pat), pick(write, pat)
    if fk is not None:
        out[name] = {"fetch_kb": round(fk, 1), "write_kb": round(wk or 0, 1), "bytes_per_launch": int(fk * 1024 * 2 + (wk or 0) * 1024)}
tf, tw = tn_split(sys.argv[1]), tn_split(sys.argv[2])
best = None
for kname, v in tf.items():                    # the grouped dW1 launch is the TN kernel with the largest fetch
    big = [x for x in v if x > 0.5 * max(v)]
    wv = tw.get(kname, [0])
    wbig = [x for x in wv if x > 0.5 * max(wv)] or [0]
    cand = {"kernel": kname.split("(")[0], "fetch_kb": round(sum(big) / len(big), 1), "write_kb": round(sum(wbig) / len(wbig), 1),
            "bytes_per_launch": int(sum(big) / len(big) * 2048 + sum(wbig) / len(wbig) * 1024), "launches_averaged": len(big)}
    if best is None or cand["fetch_kb"] > best["fetch_kb"]:
        best = cand
if best:
    out["DW1C"] = best
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
