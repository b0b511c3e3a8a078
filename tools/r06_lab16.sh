cd $GRAFT_REPO_ROOT
for pb in 3 9 3 9; do LAB_R=3 MSGAT_LAB_CCPB=$pb timeout -k 10 200 python tools/r06_lab16.py 2>&1 | grep CCPB | tail -1; done
for pb in 3 8 9 18 3; do LAB_R=5 MSGAT_LAB_CCPB=$pb timeout -k 10 200 python tools/r06_lab16.py 2>&1 | grep CCPB | tail -1; done
