cd $GRAFT_REPO_ROOT
timeout -k 10 100 build/scores_stamps_r06 > gpurun_out/stamps_n883_prio2.txt 2>&1
grep -v "id%8\|occupancy" gpurun_out/stamps_n883_prio2.txt
timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 2>&1 | tail -1
MSGAT_PARITY_LOG=gpurun_out/parity_rel_err.tsv timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
