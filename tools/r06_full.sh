cd $GRAFT_REPO_ROOT
MSGAT_PARITY_LOG=gpurun_out/parity_rel_err.tsv timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
(timeout -k 10 200 python tools/ab_pemsd4.py; cd build/ab/r05 && timeout -k 10 200 python tools/ab_pemsd4.py) 2>&1 | grep pemsd4
