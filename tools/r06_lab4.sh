cd $GRAFT_REPO_ROOT
MSGAT_DENSE_SPLIT=1 timeout -k 10 200 python tools/r06_dbg.py > gpurun_out/dbg.txt 2>&1
cat gpurun_out/dbg.txt | tail -20
