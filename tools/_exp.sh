set -e
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --heldout 0"
$B > gpurun_out/r3f_base.json 2>/dev/null
$B --preheat-ms 50 > gpurun_out/r3f_pre50.json 2>/dev/null
$B --preheat-ms 300 > gpurun_out/r3f_pre300.json 2>/dev/null
$B --preheat-ms 1500 > gpurun_out/r3f_pre1500.json 2>/dev/null
$B --warmup 40 > gpurun_out/r3f_w40.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1].ljust(22), d['ms_per_step'], d['value'], {k:v['launch_ms'] for k,v in r['other'].items()}, r['launch_ms_head'][:8], r.get('sclk_mhz'))
PY
