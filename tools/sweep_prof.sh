#!/bin/bash
# usage: tools/sweep_prof.sh "ENV=a ENV2=b" ...  -> per-GEMM timings (tools/prof_all.py) per env set, one line each
for e in "$@"; do
  echo "== $e: $(env NCX_EXPERIMENT=1 $e timeout -k 10 120 python tools/prof_all.py $PROF_FLAGS 2>/dev/null | awk '{printf "%s %s%s | ", $1, $2, ($1=="gemms"? " step " $5 : " " $(NF-2) "/" $NF)}')"
done
