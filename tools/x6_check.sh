#!/bin/bash
# GPU box: the NCX_F_X6 cases of the parity suite, then the bench line with and without --x6 (same box, back to back).
#   gpurun --timeout 900 -- 'bash tools/x6_check.sh'
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "x6 or balanced_tn or phased_backward or fused_v_gradient or fold_forms" > gpurun_out/x6_tests.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/x6_tests.log
timeout -k 10 200 python bench.py --x6 --no-cpu-baseline > gpurun_out/x6_bench.json 2> gpurun_out/x6_bench.err || { echo "bench --x6 failed"; tail -5 gpurun_out/x6_bench.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/x6_bench_default.json 2> gpurun_out/x6_bench_default.err || { echo "bench failed"; tail -5 gpurun_out/x6_bench_default.err; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/x6_bench.json", "gpurun_out/x6_bench_default.json"):
    j = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f, j["ms_per_step"], "ms", j["value"], "triplets/s", {k: v["launch_ms"] for k, v in j["roofline"]["other"].items()})
PY
