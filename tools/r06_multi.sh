cd $GRAFT_REPO_ROOT
MSGAT_BENCH_SHARE_GPU=1 timeout -k 10 500 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/bench_share2.json 2> gpurun_out/bench_share2.err
echo "share2 rc=$?"; tail -c 900 gpurun_out/bench_share2.json
MSGAT_BENCH_FORCE_DIST=1 timeout -k 10 500 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_force1.json 2> gpurun_out/bench_force1.err
echo "force1 rc=$?"; tail -c 600 gpurun_out/bench_force1.json
