set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "configs4 or bf16" 2>&1 | tail -2
bash tools/kstats.sh r3j --bf16 --K 48 --batch 1024 2>&1 | tail -24
