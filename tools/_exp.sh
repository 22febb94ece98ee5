set -e
cd $GRAFT_REPO_ROOT
cp profiles/r3_traffic.json gpurun_out/r3_traffic.json
bash tools/profile_round.sh r3_02_c5 c5 --bf16 --K 48 --batch 1024 > gpurun_out/pr_c5.log 2>&1
cp gpurun_out/r3_traffic.json profiles/r3_traffic.json
python bench.py --bf16 --K 48 --batch 1024 --no-cpu-baseline > gpurun_out/r3_02_c5_bench.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_02_c5_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], 'frac', r['frac'], 'traffic', r['traffic'], {k:(v['launch_ms'],v['tflops']) for k,v in r['other'].items()})
t=json.load(open('gpurun_out/r3_traffic.json'))['c5']
print({k:v.get('bytes_per_launch') for k,v in t.items() if isinstance(v,dict)})
PY
