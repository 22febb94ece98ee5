#!/bin/bash
# Quick per-kernel times of the default bench under rocprofv3 (kernel trace only):  tools/kstats.sh <tag> [bench args]
TAG=${1:-k}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --heldout 0 "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.err
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
python3 - $OUT/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r['Name'][:80].ljust(80), r['Calls'].rjust(4), "%9.1f" % (float(r['AverageNs']) / 1e3))
PY
tail -1 $OUT/${TAG}_bench.json | cut -c1-200
rm -rf $OUT/${TAG}_trace
