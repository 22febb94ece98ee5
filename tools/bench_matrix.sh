#!/bin/bash
# GPU box: the workload table of DESIGN section 7 on the current build (no CPU baseline): one bench line per workload into gpurun_out/<tag>_matrix.json
TAG=${1:-r4_07}
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/mx_$name.json 2> gpurun_out/mx_$name.err || echo "FAILED $name"; }
run c2
run c2_x6 --x6
run c3 --c3
run c3_x6 --c3 --x6
run c5 --bf16 --K 48 --batch 1024
run c5_fp32 --K 48 --batch 1024
run b64 --scaling strong --batch 64
run b128 --scaling strong --batch 128
run b256 --scaling strong --batch 256
run b2048 --batch 2048
run h1024l2 --H 1024 --L 2
run h512l3 --H 512 --L 3
python - "$TAG" <<'PY'
import json, sys, glob, os
out = {}
for f in sorted(glob.glob("gpurun_out/mx_*.json")):
    n = os.path.basename(f)[3:-5]
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = j.get("roofline") or {}
        out[n] = dict(triplets_per_s=j["value"], ms_per_step=j["ms_per_step"], workload=j["config"]["workload"][:90], dtype=j["dtype"][:40],
                      main_ms=(r.get("other", {}).get("MAIN") or {}).get("launch_ms"), dw1c_ms=(r.get("other", {}).get("DW1C") or {}).get("launch_ms"))
        print(n, j["value"], j["ms_per_step"], out[n]["main_ms"], out[n]["dw1c_ms"])
    except Exception as e:
        print(n, "ERR", e)
json.dump(out, open("gpurun_out/%s_matrix.json" % sys.argv[1], "w"), indent=1)
PY
