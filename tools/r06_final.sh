cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_driver_args.json 2> gpurun_out/final/bench_driver_args.err; echo "bench1 rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/final/bench_plain.json 2> gpurun_out/final/bench_plain.err; echo "bench2 rc=$?"
(timeout -k 10 200 python tools/ab_step.py; cd build/ab/r05 && timeout -k 10 200 python tools/ab_step.py; cd $GRAFT_REPO_ROOT; timeout -k 10 200 python tools/ab_step.py; cd build/ab/r05 && timeout -k 10 200 python tools/ab_step.py) 2>&1 | grep "hot-path" > gpurun_out/final/ab_r05_vs_r06.txt
cat gpurun_out/final/ab_r05_vs_r06.txt
tail -c 700 gpurun_out/final/bench_driver_args.json
