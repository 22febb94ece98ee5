#!/bin/bash
# GPU box: five invocations of the default bench line back to back (no CPU baseline), min / median / max into gpurun_out/<tag>_bench_repeats.json
TAG=${1:-r4_06}
mkdir -p gpurun_out
for i in 1 2 3 4 5; do timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/rep_$i.json 2> gpurun_out/rep_$i.err || exit 1; done
python - "$TAG" <<'PY'
import json, sys
rows = []
for i in range(1, 6):
    j = json.loads([l for l in open("gpurun_out/rep_%d.json" % i) if l.startswith("{")][-1])
    r = j["roofline"]
    rows.append(dict(ms_per_step=j["ms_per_step"], value=j["value"], ms_per_step_no_preheat=j["ms_per_step_no_preheat"]["ms_per_step"], main_ms=r["launch_ms"], frac=r["frac"],
                     frac_executed=r.get("frac_executed"), dw1c_ms=r["other"]["DW1C"]["launch_ms"], sclk_mhz=r.get("sclk_mhz")))
v = sorted(x["value"] for x in rows)
out = dict(what="five invocations of `python bench.py --no-cpu-baseline` back to back on one box", triplets_per_s_min=v[0], triplets_per_s_median=v[2], triplets_per_s_max=v[4], runs=rows)
json.dump(out, open("gpurun_out/%s_bench_repeats.json" % sys.argv[1], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("triplets_per_s_min", "triplets_per_s_median", "triplets_per_s_max")}), [x["ms_per_step"] for x in rows], [x["main_ms"] for x in rows], [x["dw1c_ms"] for x in rows])
PY
