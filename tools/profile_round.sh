#!/bin/bash
# One round's judged evidence, on the GPU box:  tools/profile_round.sh r2   (writes gpurun_out/<tag>_*)
# kernel trace + stats of the default bench, then the PMC passes the MI355X guide prescribes (separate runs, --kernel-trace only).
set -e
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --heldout 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $B > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.err
S="python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --heldout 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $S > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $S > /dev/null 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_sq1 -- $S > /dev/null 2> $OUT/${TAG}_pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_sq2 -- $S > /dev/null 2> $OUT/${TAG}_pmc_sq2.err
python3 $ROOT/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_traffic.json > /dev/null
python3 $ROOT/tools/pmc_mfma.py $OUT/${TAG}_pmc_sq1 $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_mfma.txt > /dev/null
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
echo done
