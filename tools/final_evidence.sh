#!/bin/bash
# GPU box: the round's final evidence on ONE box -- the complete default bench line, its rocprofv3 / PMC summaries, then the --x6 line (never the headline).
#   gpurun --timeout 1150 -- 'bash tools/final_evidence.sh r4_03'
TAG=${1:-r4_03}
mkdir -p gpurun_out
cp profiles/r4_traffic.json gpurun_out/r4_traffic.json 2>/dev/null
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
timeout -k 10 300 python bench.py --x6 > gpurun_out/r4_x6_bench.json 2> gpurun_out/r4_x6_bench.err || { echo "bench --x6 failed"; tail -5 gpurun_out/r4_x6_bench.err; exit 1; }
timeout -k 10 600 bash tools/profile_round.sh ${TAG} c2 || { echo "profile_round failed"; exit 1; }
python - <<PY
import csv, json
for f in ("gpurun_out/${TAG}_bench.json", "gpurun_out/r4_x6_bench.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    r = j["roofline"]
    print(f, j["ms_per_step"], "ms", j["value"], "triplets/s; cold", j.get("ms_per_step_no_preheat", {}).get("ms_per_step"), "frac", r["frac"], r.get("frac_executed"), "traffic", r["traffic"],
          {k: v["launch_ms"] for k, v in r["other"].items()}, "cpu", j.get("cpu_baseline", {}).get("value"))
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
