#!/bin/bash
# usage: tools/sweep2.sh "ENV1=a ENV2=b" "ENV1=c" ...   -> bench line summary per env set (extra bench flags in $BENCH_FLAGS)
for e in "$@"; do
  env NCX_EXPERIMENT=1 $e timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline $BENCH_FLAGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], {k:(v['launch_ms'],v['plan']['tile'],v['plan']['ksplit']) for k,v in d['roofline']['other'].items()})"
done
