cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python3 tools/host_overhead.py --workload pemsd7 --dropin --steps 200 2>&1 | cut -c1-150 > gpurun_out/r05/host_overhead_dropin_before.txt
python3 - <<'PY' > gpurun_out/r05/dropin_before.json 2>gpurun_out/r05/dropin_before.err
import json, torch, bench
dev = torch.device("cuda:0")
hp = bench.HotPath(bench.WORKLOADS["pemsd7"], dev, 0)
print(json.dumps(bench.dropin_object(hp, dev), indent=1))
PY
cat gpurun_out/r05/dropin_before.json; tail -5 gpurun_out/r05/dropin_before.err; head -60 gpurun_out/r05/host_overhead_dropin_before.txt
