cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-baselines --steps 20 --warmup 5 > $O/bench_hot_under_rocprof.json 2>/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stress_kt -- python3 $R/tools/stress_kernels.py --reps 3 > $O/stress_kernels.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/stress_fetch -- python3 $R/tools/stress_kernels.py --reps 2 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/stress_write -- python3 $R/tools/stress_kernels.py --reps 2 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/hot_fetch -- python3 $R/tools/kbench.py --eager --reps 5 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/hot_write -- python3 $R/tools/kbench.py --eager --reps 5 > /dev/null 2>&1
python3 $R/tools/stress_kernels.py --reps 5 2>&1 | grep -v "^W2026\|amdgpu" > $O/stress_kernels_unprofiled.log
echo done
