#!/bin/bash
# GPU box: the dE product (k_main_fwd, two plain segments, 16 k-steps) on other tile shapes through the experiment hook NCX_MAIN_CFG (kernel times under rocprofv3).
cd /tmp && export TMPDIR=/tmp
for c in none 3 2 0; do
  if [ $c = none ]; then unset NCX_EXPERIMENT NCX_MAIN_CFG; else export NCX_EXPERIMENT=1 NCX_MAIN_CFG=$c; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --heldout 0 > /tmp/b_$c.json 2>/dev/null
  f=$(find /tmp/tr_$c -name "*kernel_stats.csv")
  python3 - "$f" "$c" /tmp/b_$c.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print("cfg", sys.argv[2], "step", j["ms_per_step"], [(r["Name"][22:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows if "k_main_fwd" in r["Name"]])
PY
done
