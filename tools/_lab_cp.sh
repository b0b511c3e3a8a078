cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2 && cd /tmp && \
for r in 3 5; do rm -rf $O/kt$r; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$r -- python3 $R/tools/full_model_profile.py --R $r --steps 6 > $O/run$r.log 2>&1; tail -1 $O/run$r.log; python3 $R/tools/trace_summary.py $O/kt$r/*/*kernel_trace.csv k_adam_advance 1 3 60 > $O/summary_R$r.txt; head -8 $O/summary_R$r.txt; done
