set -e
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --heldout 0"
for v in A B A B; do NCX_LIB=$PWD/vqa-counterexamples_amd/lib/lib$v.so $B > gpurun_out/r3h_$v.json 2>/dev/null; python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3h_%s.json'%sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], d['value'], {k:v['launch_ms'] for k,v in r['other'].items()}, r.get('sclk_mhz'))
PY
done
NCX_LIB=$PWD/vqa-counterexamples_amd/lib/libB.so python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "configs1 or ragged or golden" 2>&1 | tail -2
