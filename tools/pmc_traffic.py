"""Parse two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs with --kernel-trace only, as
MI355X_MICROARCH.md prescribes) of `bench.py --steps 4 --warmup 2 [workload flags]` into HBM bytes per launch:
bytes = FETCH_SIZE[KB] * 1024 * 2 (gfx950 tallies a 128-B read request as 64 B) + WRITE_SIZE[KB] * 1024.
bench.py's kernel ids: MAIN = the forward product of linear_1 (one launch); DW1C = the linear_1 weight gradient, the SUM of the
launches bench.py times under that id (fp32: k_dw_km + grouped TN GEMM + their merged reduction; bf16: cast + TN GEMM + reduce).
k_prep / k_adam are listed as calibration against known byte counts.
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [workload key: c2 | c3 | c5] [source note]"""
import csv, glob, json, os, sys
from collections import defaultdict


def by_kernel(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = by_kernel(sys.argv[1], "FETCH_SIZE"), by_kernel(sys.argv[2], "WRITE_SIZE")
key = sys.argv[4] if len(sys.argv) > 4 else "c2"


def big(vals):
    """mean over the launches of the big problem of a kernel name shared by several problems (those above half the maximum)"""
    if not vals:
        return 0.0, 0
    b = [x for x in vals if x > 0.5 * max(vals)] if max(vals) > 0 else vals
    return sum(b) / len(b), len(b)


def part(pat, not_pat=()):
    names = [k for k in fetch if all(p in k for p in pat) and not any(p in k for p in not_pat)]
    if not names:
        return None
    k = max(names, key=lambda n: max(fetch[n]))                 # the instantiation that moves the most
    f, nf = big(fetch[k]); w, _ = big(write.get(k, [0.0]))
    return {"kernel": k.split("(")[0][:90], "fetch_kb": round(f, 1), "write_kb": round(w, 1), "bytes": int(f * 2048 + w * 1024), "launches_averaged": nf}


def total(parts):
    parts = [p for p in parts if p]
    return {"bytes_per_launch": sum(p["bytes"] for p in parts), "parts": parts} if parts else None


out = {}
bf16 = any("gemm_bf16_nt" in k for k in fetch) and key == "c5"
if bf16:
    out["MAIN"] = total([part(("gemm_bf16_nt8",)) or part(("gemm_bf16_nt_kernel<128",))])
    out["DW1C"] = total([part(("k_dpre_to_bf16",)), part(("gemm_bf16_tn8_kernel",)) or part(("gemm_bf16_tn_kernel",)), part(("k_bf16_reduce_dwc",))])
else:
    out["MAIN"] = total([part(("k_main_fwd", "MainCfg<96, 64")) or part(("k_main_fwd",))])
    # round 4: the fold kernel + the balanced 8-wave TN launch (ncx_dwtn.hip) + their merged reduction; earlier builds: the grouped TN GEMM
    out["DW1C"] = total([part(("k_dw_km_x6",)) or part(("k_dw_km8",)) or part(("k_dw_km<",)), part(("k_dw_tn8",)) or part(("seg_gemm_kernel<128, 64, false, false",)),
                         part(("k_dw_reduce_km_tn8",)) or part(("k_dw_km_reduce_fixup",))])
for name, pat in (("k_prep", ("k_prep",)), ("k_adam", ("k_adam",))):
    p = part(pat)
    if p:
        out[name] = {"bytes_per_launch": p["bytes"], "fetch_kb": p["fetch_kb"], "write_kb": p["write_kb"]}
out["_source"] = sys.argv[5] if len(sys.argv) > 5 else "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (separate runs, --kernel-trace only)"
allj = json.load(open(sys.argv[3])) if os.path.exists(sys.argv[3]) else {
    "_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on bench.py --steps 4 --warmup 2 --no-cpu-baseline "
            "[workload flags]; bytes per launch = FETCH_SIZE[KB]*1024*2 (gfx950 counts a 128-B read request as 64 B: MI355X_MICROARCH.md, HBM) "
            "+ WRITE_SIZE[KB]*1024; DW1C = sum over the launches bench.py times under that id"}
allj[key] = out
json.dump(allj, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
