cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/parity_r06.log 2>&1
echo "parity rc=$?"
tail -5 gpurun_out/parity_r06.log
timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 > gpurun_out/dense_pemsd7.txt 2>&1; tail -1 gpurun_out/dense_pemsd7.txt
(timeout -k 10 200 python tools/ab_step.py; cd build/ab/r05 && timeout -k 10 200 python tools/ab_step.py; cd $GRAFT_REPO_ROOT; timeout -k 10 200 python tools/ab_step.py; cd build/ab/r05 && timeout -k 10 200 python tools/ab_step.py) > gpurun_out/ab.txt 2>&1
grep "hot-path" gpurun_out/ab.txt
