cd $GRAFT_REPO_ROOT
export MSGAT_DENSE_BF16=1
for v in "" build/lab/libmsgat_blab1.so build/lab/libmsgat_blab2.so build/lab/libmsgat_blab4.so build/lab/libmsgat_blab7.so; do
  if [ -z "$v" ]; then a=""; else a="--lib $v"; fi
  timeout -k 10 200 python tools/dense_bench.py --workload stress $a 2>&1 | tail -1
  timeout -k 10 200 python tools/dense_bench.py --workload pemsd7 $a 2>&1 | tail -1
done > gpurun_out/blab.txt 2>&1
cat gpurun_out/blab.txt
