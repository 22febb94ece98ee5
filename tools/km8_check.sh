#!/bin/bash
# GPU box: k_dw_km8 (and its two-steps-per-barrier variant, hook NCX_KM8X2) against the oracle on forced shapes, then the default bench line with each form on the same box.
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "8_wave_form or configs1_full_size" > gpurun_out/km8_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/km8_tests.log
NCX_KM8X2=1 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "8_wave_form" > gpurun_out/km8x2_tests.log 2>&1; echo "pytest x2 rc $?"; tail -3 gpurun_out/km8x2_tests.log
NCX_EXPERIMENT=1 NCX_KM8X2=1 timeout -k 10 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "configs1_full_size" > gpurun_out/km8x2_tests2.log 2>&1; echo "pytest x2 full rc $?"; tail -3 gpurun_out/km8x2_tests2.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/km8_bench.json 2> gpurun_out/km8_bench.err || { echo "bench km8 failed"; tail -5 gpurun_out/km8_bench.err; exit 1; }
NCX_EXPERIMENT=1 NCX_NO_KM8=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/km8_bench_default.json 2> gpurun_out/km8_bench_default.err
NCX_EXPERIMENT=1 NCX_KM8X2=1 timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/km8_bench2.json 2> gpurun_out/km8_bench2.err
python - <<'PY'
import json
for f in ("gpurun_out/km8_bench.json", "gpurun_out/km8_bench_default.json", "gpurun_out/km8_bench2.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, j["ms_per_step"], "ms", j["value"], "triplets/s", {k: v["launch_ms"] for k, v in j["roofline"]["other"].items()})
PY
