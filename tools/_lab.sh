cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in 0 1 2 3 4; do echo "LAB=$v"; MSGAT_AGG_LAB=$v timeout -k 10 120 python3 $R/tools/stress_kernels.py --reps 3 2>&1 | grep "^aggregate"; done
