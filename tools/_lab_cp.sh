cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/step; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 && cd /tmp && \
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --no-baselines --steps 20 --warmup 5 > $O/bench.json 2>/dev/null
python3 $R/tools/trace_one_step.py $O/kt --per-step 24 --skip 89
