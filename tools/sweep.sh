#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   -> bench line summary per value
VAR=$1; shift
for v in "$@"; do
  env NCX_EXPERIMENT=1 $VAR=$v timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], {k:(v['launch_ms'],v['plan']['ksplit']) for k,v in d['roofline']['other'].items()})"
done
