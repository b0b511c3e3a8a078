cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python -m pytest $R/tests/test_gpu_branches.py $R/tests/test_gpu_model.py -x -q 2>&1 | tail -2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/full3 -- python3 $R/tools/full_model_profile.py --R 3 --steps 6 > $O/full3.log 2>&1
python3 $R/tools/trace_one_step.py $O/full3 --anchor k_adam_advance > $O/train_step_launches.txt 2>&1
python3 $R/tools/trace_summary.py $O/full3/*/*kernel_trace.csv k_adam_advance 1 3 70 > $O/full_step_R3_kernels.txt
rm -rf $O/full3
grep -c "" $O/train_step_launches.txt; grep "k_reduce\|busy" $O/train_step_launches.txt
cd $R; for args in "--R 3" "--R 3 --graph"; do python3 tools/full_model_profile.py $args --steps 10 2>/dev/null | tail -1; done
