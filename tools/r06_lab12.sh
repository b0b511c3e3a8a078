cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_branches.py tests/test_gpu_layernorm.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
(timeout -k 10 300 python tools/ab_step.py; cd build/ab/r05 && timeout -k 10 300 python tools/ab_step.py; cd $GRAFT_REPO_ROOT; timeout -k 10 300 python tools/ab_step.py) 2>&1 | grep "hot-path"
