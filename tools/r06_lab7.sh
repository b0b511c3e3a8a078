cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 2>&1 | tail -1
timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 --lib build/lab/libmsgat_slab.so 2>&1 | tail -1
