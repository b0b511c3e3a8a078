cd $GRAFT_REPO_ROOT
for l in "" "--lib build/lab/libmsgat_aux2.so" "--lib build/lab/libmsgat_aux3.so" ""; do
  timeout -k 10 120 python tools/kbench.py --only aggregate --sets 4 --reps 40 $l 2>&1 | tail -1
  timeout -k 10 120 python tools/kbench.py --only aggregate --sets 1 --reps 40 $l 2>&1 | tail -1
done
