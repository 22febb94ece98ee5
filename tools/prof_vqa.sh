#!/bin/bash
# usage: tools/prof_vqa.sh  -> per-kernel avg us of ncx kernels in tools/bench_vqa.py under current env
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python3 /root/repo/tools/bench_vqa.py > /tmp/pv.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pv/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ncx::seg_gemm' in r['Name']: print("   %-70s calls %4s avg %8.1f us" % (r['Name'][10:80], r['Calls'], float(r['AverageNs'])/1e3))
PY
