cd $GRAFT_REPO_ROOT
MSGAT_PARITY_LOG=gpurun_out/parity_rel_err.tsv timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench1.json 2> gpurun_out/bench1.err
echo "bench rc=$?"; tail -c 1500 gpurun_out/bench1.json
