cd $GRAFT_REPO_ROOT
for pb in 3 4 6 9 3; do MSGAT_LAB_CCPB=$pb timeout -k 10 200 python tools/r06_lab16.py 2>&1 | grep CCPB; done
