cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-baselines 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run', d['ms_per_step'], d['ms_per_step_median_hip_events'])"; done
