cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tail.py -q -x -m gpu -k "leading_axis or drop_in or autocast or golden" 2>&1 | tail -3
python3 tools/host_overhead.py --workload pemsd7 --dropin --steps 200 2>&1 | cut -c1-150 > gpurun_out/r05/host_overhead_dropin_after.txt
python3 - <<'PY' > gpurun_out/r05/dropin_after.json 2>gpurun_out/r05/dropin_after.err
import json, torch, bench
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS["pemsd7"], dev, 0)
print(json.dumps(bench.dropin_object(hp, dev), indent=1))
PY
cat gpurun_out/r05/dropin_after.json; head -45 gpurun_out/r05/host_overhead_dropin_after.txt
python3 tools/train_kernels.py --R 3 > gpurun_out/r05/train_kernels_stacked.txt 2>/dev/null
python3 tools/train_kernels.py --R 3 --unstacked > gpurun_out/r05/train_kernels_unstacked.txt 2>/dev/null
head -30 gpurun_out/r05/train_kernels_unstacked.txt
