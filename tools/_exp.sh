set -e
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --heldout 0"
for m in 0 2 0 2; do NCX_SIDE_STREAM=$m $B > gpurun_out/r3k_$m.json 2>/dev/null; python - $m <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3k_%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], d['value'], {k:v['launch_ms'] for k,v in r['other'].items()})
PY
done
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "192x256" 2>&1 | tail -3
