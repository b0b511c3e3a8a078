cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; mkdir -p $O
cd $R && timeout -k 10 900 python3 -m pytest tests/test_gpu_branches.py tests/test_gpu_model.py -x -q 2>&1 | tail -2 && cd /tmp && \
rm -rf $O/kt3; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt3 -- python3 $R/tools/full_model_profile.py --R 3 --steps 6 > $O/run3.log 2>&1; tail -1 $O/run3.log; python3 $R/tools/trace_summary.py $O/kt3/*/*kernel_trace.csv k_adam_advance 1 3 60 | grep "steps:\|tmix"
timeout -k 10 200 python3 $R/tools/full_model_profile.py --R 5 --steps 10 2>/dev/null | tail -1
