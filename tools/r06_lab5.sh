cd $GRAFT_REPO_ROOT
for v in h f; do timeout -k 10 120 python tools/r06_dbg2.py $v > gpurun_out/dbg_$v.log 2>&1; echo "$v rc=$?"; grep -v "Extension\|^$" gpurun_out/dbg_$v.log | head -6; done
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_tail.py tests/test_gpu_multirank.py -x -q > gpurun_out/t4.log 2>&1; echo "model tests rc=$?"; tail -4 gpurun_out/t4.log
