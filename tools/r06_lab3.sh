cd $GRAFT_REPO_ROOT
export MSGAT_DENSE_BF16=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dense -o dense -- python3 $GRAFT_REPO_ROOT/tools/dense_bench.py --workload pemsd7 > $GRAFT_REPO_ROOT/gpurun_out/prof_dense.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_dense -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_dense -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-200
