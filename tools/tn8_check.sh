#!/bin/bash
# GPU box: the balanced TN launch after a loader change -- its parity cases, the phased / pipelined bit-identity tests, the full-size case; then the default bench line.
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "balanced_tn or phased or configs1_full_size or configs4 or pipelined or lesion or golden" > gpurun_out/tn8_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/tn8_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/tn8_bench.json 2> gpurun_out/tn8_bench.err || { echo "bench failed"; tail -5 gpurun_out/tn8_bench.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/tn8_bench2.json 2> gpurun_out/tn8_bench2.err
python - <<'PY'
import json
for f in ("gpurun_out/tn8_bench.json", "gpurun_out/tn8_bench2.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, j["ms_per_step"], "ms", j["value"], "triplets/s", {k: v["launch_ms"] for k, v in j["roofline"]["other"].items()})
PY
