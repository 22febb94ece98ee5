#!/bin/bash
# GPU box: k_dw_km8 against the oracle on forced shapes, then the default bench line with and without it (hook NCX_NO_KM8) on the same box.
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "8_wave_form or configs1_full_size" > gpurun_out/km8_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/km8_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/km8_bench.json 2> gpurun_out/km8_bench.err || { echo "bench km8 failed"; tail -5 gpurun_out/km8_bench.err; exit 1; }
NCX_EXPERIMENT=1 NCX_NO_KM8=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/km8_bench_default.json 2> gpurun_out/km8_bench_default.err
python - <<'PY'
import json
for f in ("gpurun_out/km8_bench.json", "gpurun_out/km8_bench_default.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, j["ms_per_step"], "ms", j["value"], "triplets/s", {k: v["launch_ms"] for k, v in j["roofline"]["other"].items()})
PY
