#!/bin/bash
# GPU box: the forward kernel after a loader change -- fold forms (bit-identity), goldens, full-size cases, stamps; then the default bench line twice.
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "fold_forms or configs1 or golden or ragged or distance_inside or phased_forward or clock_stamps or x6_forward" > gpurun_out/main_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/main_tests.log
timeout -k 10 200 python tools/main_stamps.py > gpurun_out/main_stamps.log 2>&1; tail -6 gpurun_out/main_stamps.log
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/main_bench$i.json 2> gpurun_out/main_bench$i.err; done
python - <<'PY'
import json
for f in ("gpurun_out/main_bench1.json", "gpurun_out/main_bench2.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, j["ms_per_step"], "ms", j["value"], "triplets/s", {k: v["launch_ms"] for k, v in j["roofline"]["other"].items()}, "frac", j["roofline"]["frac"])
PY
