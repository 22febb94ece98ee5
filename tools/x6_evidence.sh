#!/bin/bash
# GPU box: the judged evidence of the NCX_F_X6 line (never the headline): the full bench line, then kernel stats + PMC traffic under rocprofv3.
#   gpurun --timeout 1100 -- 'bash tools/x6_evidence.sh'
mkdir -p gpurun_out
cp profiles/r4_traffic.json gpurun_out/r4_traffic.json 2>/dev/null
timeout -k 10 300 python bench.py --x6 > gpurun_out/r4_x6_bench.json 2> gpurun_out/r4_x6_bench.err || { echo "bench --x6 failed"; tail -5 gpurun_out/r4_x6_bench.err; exit 1; }
timeout -k 10 600 bash tools/profile_round.sh r4_x6 x6 --x6 || { echo "profile_round failed"; exit 1; }
python - <<'PY'
import csv, json
j = json.loads([l for l in open("gpurun_out/r4_x6_bench.json") if l.startswith("{")][-1])
print(j["ms_per_step"], "ms", j["value"], "triplets/s", {k: v["launch_ms"] for k, v in j["roofline"]["other"].items()})
rows = list(csv.DictReader(open("gpurun_out/r4_x6_kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
