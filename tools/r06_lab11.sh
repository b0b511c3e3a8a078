cd $GRAFT_REPO_ROOT
(timeout -k 10 200 python tools/ab_pemsd4.py; cd build/ab/r05 && timeout -k 10 200 python tools/ab_pemsd4.py; cd $GRAFT_REPO_ROOT; timeout -k 10 200 python tools/ab_pemsd4.py) 2>&1 | grep pemsd4
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
