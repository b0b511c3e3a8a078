cd $GRAFT_REPO_ROOT
echo "== kbench --sets 4 (cold operands) =="
timeout -k 10 300 python tools/kbench.py --sets 4 --only mix_fwd,seg_fwd98 2>&1 | tail -8
echo "== causal conv, cold =="
timeout -k 10 300 python tools/causal_conv_time.py 2>&1 | tail -8
echo "== causal conv, warm =="
timeout -k 10 300 python tools/causal_conv_time.py --warm 2>&1 | tail -8
