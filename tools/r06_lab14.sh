cd $GRAFT_REPO_ROOT
for cc in 0 1 2 3 4 7; do for pb in 3 6; do
echo "== MSGAT_LAB_CC=$cc MSGAT_LAB_CCPB=$pb (warm) =="
MSGAT_LAB_CC=$cc MSGAT_LAB_CCPB=$pb timeout -k 10 200 python tools/causal_conv_time.py --warm --lib build/lab/libmsgat_lab.so 2>&1 | grep "Cr=24 Co=24 N=883"
done; done
