set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r3_01 c2 > gpurun_out/pr_c2.log 2>&1
bash tools/profile_round.sh r3_01_c3 c3 --c3 > gpurun_out/pr_c3.log 2>&1
bash tools/profile_round.sh r3_01_c5 c5 --bf16 --K 48 --batch 1024 > gpurun_out/pr_c5.log 2>&1
cp gpurun_out/r3_traffic.json profiles/r3_traffic.json
python bench.py > gpurun_out/r3_01_bench.json 2> gpurun_out/r3_01_bench.err
python bench.py --c3 --no-cpu-baseline > gpurun_out/r3_01_c3_bench.json 2>/dev/null
python bench.py --bf16 --K 48 --batch 1024 --no-cpu-baseline > gpurun_out/r3_01_c5_bench.json 2>/dev/null
python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3_01_bench_100.json 2>/dev/null
python bench.py --scaling strong --batch 64 --no-cpu-baseline --heldout 0 > gpurun_out/r3_01_b64.json 2>/dev/null
tail -c 600 gpurun_out/r3_01_bench.json
