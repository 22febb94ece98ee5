#!/bin/bash
# per-dispatch durations of the phased backward's kernels (tools/exp_dw1c_split.py) on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/exp_split_trace -- python3 $ROOT/tools/exp_dw1c_split.py 12 > $OUT/exp_split.log 2> $OUT/exp_split.err
python3 - $OUT/exp_split_trace <<'PY' >> $OUT/exp_split.log
import csv, glob, sys
from collections import defaultdict
rows = []
for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "ncx::" in n:
        seq[n.split("(")[0][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in seq.items():
    tail = v[-24:]
    print("%-72s n=%3d  last: %s" % (n, len(v), " ".join("%.1f" % x for x in tail[-12:])))
PY
rm -rf $OUT/exp_split_trace
tail -30 $OUT/exp_split.log
