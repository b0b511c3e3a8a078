mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_second.json 2> gpurun_out/r05/bench_second.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05/bench_second.json"))
print(d["ms_per_step"], d["ms_per_step_median_hip_events"], d["roofline"]["frac"])
for k in ("full_step_cfg3","full_step_cfg4"):
    print(k, {a:b for a,b in d[k].items() if "ms" in a})
print({k:v for k,v in d["dropin_loop"].items() if "train" in k})
print(d["widths48"]["train_step"]["ms_per_step"], d["widths96"]["train_step"]["ms_per_step"], d["stress"]["ms_per_step"], d["pemsd4"]["ms_per_step"], d["pemsd4"]["train_step_eager_launch_ms"], d["pemsd4"]["train_step_hip_graph_ms"])
PY
