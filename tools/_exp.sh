set -e
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --heldout 0"
for m in "0 0" "0 1" "3 1" "1 1" "2 1"; do set -- $m; env NCX_SIDE_STREAM=$1 NCX_BENCH_HIPRIO=$( [ $2 = 1 ] && echo 1 ) $B > gpurun_out/r3l.json 2>/dev/null; python - "$m" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r3l.json').read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['ms_per_step'], d['value'], {k:v['launch_ms'] for k,v in r['other'].items()})
PY
done
