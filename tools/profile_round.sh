#!/bin/bash
# One round's judged evidence, on the GPU box:  tools/profile_round.sh <tag> <workload key: c2|c3|c5> [bench flags of the workload]
#   c2 (BASELINE configs[1], the headline): tools/profile_round.sh r4_01 c2
#   c3 (configs[2], + fused MUTAN producer): tools/profile_round.sh r4_c3 c3 --c3
#   c5 (configs[4], K=48 B=1024 bf16):       tools/profile_round.sh r4_c5 c5 --bf16 --K 48 --batch 1024
#   x6 (configs[1] under NCX_F_X6):          tools/profile_round.sh r4_x6 x6 --x6
# kernel trace + stats of the bench, then the PMC passes the MI355X guide prescribes (separate runs, --kernel-trace only).
set -e
TAG=${1:-r4}; KEY=${2:-c2}
if [ $# -ge 2 ]; then shift 2; else shift $#; fi
RND=${TAG%%[!a-z0-9]*}; RND=${RND:0:2}              # round prefix of the tag (r4_01 -> r4): the traffic file is per round
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --heldout 0 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $B > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.err
S="python3 $ROOT/bench.py --steps 4 --warmup 2 --preheat-ms 0 --no-cpu-baseline --heldout 0 $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $S > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- $S > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 $ROOT/tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${RND}_traffic.json $KEY "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh $TAG $KEY $*)" > /dev/null
if [ "$KEY" = "c2" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_sq1 -- $S > /dev/null 2> $OUT/${TAG}_pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_sq2 -- $S > /dev/null 2> $OUT/${TAG}_pmc_sq2.err
python3 $ROOT/tools/pmc_mfma.py $OUT/${TAG}_pmc_sq1 $OUT/${TAG}_pmc_sq2 $OUT/${TAG}_pmc_mfma.txt > /dev/null
fi
find $OUT/${TAG}_trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
# raw per-dispatch counters are large: keep only the per-kernel means the tools extracted
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_sq1 $OUT/${TAG}_pmc_sq2
echo done $TAG
