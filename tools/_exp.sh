set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dropin_gpu.py -x -q -m gpu -k "vqa or mutan or golden" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --heldout 0"
for v in 0 1; do env NCX_EXPERIMENT=1 $( [ $v = 1 ] && echo NCX_VQA_NO_MAIN=1 ) $B --c3 > gpurun_out/r3o.json 2>/dev/null; python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3o.json').read().strip().splitlines()[-1]); r=d['roofline']
print('no_main' if sys.argv[1]=='1' else 'main', d['ms_per_step'], d['value'])
PY
done
$B > gpurun_out/r3o_c2.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3o_c2.json').read().strip().splitlines()[-1]); r=d['roofline']
print('c2', d['ms_per_step'], d['value'], {k:v['launch_ms'] for k,v in r['other'].items()})
PY
