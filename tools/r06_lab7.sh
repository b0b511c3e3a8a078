cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 2>&1 | tail -1
MSGAT_SCORES7S=0 timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 2>&1 | tail -1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_outputs or headline or helper_wave or pemsd7" 2>&1 | tail -5
