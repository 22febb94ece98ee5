set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --heldout 0"
$B > gpurun_out/r3g_pipe.json 2>/dev/null
$B > gpurun_out/r3g_pipe2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(22), d['ms_per_step'], d['value'], {k:v['launch_ms'] for k,v in r['other'].items()}, r.get('sclk_mhz'))
PY
bash tools/kstats.sh r3g 2>&1 | tail -18
