set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "192x256 or configs4 or bf16" 2>&1 | tail -3
bash tools/kstats.sh r3n --bf16 --K 48 --batch 1024 2>&1 | tail -20
