#!/bin/bash
# End-of-round measurements (GPU box): kernel traces of the hot path and the training steps, per-kernel tables of the three
# registry models, the bench line with the driver's arguments and with the defaults, the five-rank rehearsal on one GPU.
#     gpurun --timeout 1200 -- 'bash tools/final_measurements.sh r06'   (results under gpurun_out/<round>/)
RND=${1:-r06}
mkdir -p gpurun_out/$RND
bash tools/profile_round.sh $RND "hot full" > gpurun_out/profile_round_${RND}b.log 2>&1
python3 tools/train_kernels.py --R 3 --top 60 > gpurun_out/$RND/train_kernels_stacked_final.txt 2>/dev/null
python3 tools/train_kernels.py --R 3 --hidden 48 --top 30 > gpurun_out/$RND/train_kernels_msgat48.txt 2>/dev/null
python3 tools/train_kernels.py --R 3 --hidden 96 --top 30 > gpurun_out/$RND/train_kernels_msgat96.txt 2>/dev/null
python3 tools/train_kernels.py --R 3 --ops --top 0 2>/dev/null | grep -E "aten::|torch operators" > gpurun_out/$RND/train_step_torch_ops.txt
python3 tools/host_overhead_train.py --small > gpurun_out/$RND/host_overhead_train_small.txt 2>&1
python3 tools/causal_conv_time.py > gpurun_out/$RND/causal_conv_cold.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/$RND/bench_driver_args.json 2> gpurun_out/$RND/bench_driver_args.err
python bench.py > gpurun_out/$RND/bench_plain.json 2> gpurun_out/$RND/bench_plain.err
MSGAT_BENCH_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 5 --steps 10 --warmup 3 > gpurun_out/$RND/bench_share_gpu_5ranks.json 2> gpurun_out/$RND/bench_share_gpu_5ranks.err
RND=$RND python - <<PY
import json, os
RND = os.environ["RND"]
for f in ("bench_driver_args","bench_plain","bench_share_gpu_5ranks"):
    try:
        d=json.load(open(f"gpurun_out/{RND}/{f}.json"))
        print(f, d["n_gpus"], d["ms_per_step"], d["ms_per_step_median_hip_events"], d["roofline"]["frac"], d.get("full_step_cfg3",{}).get("ms_per_step"), d.get("full_step_cfg3",{}).get("ms_per_step_median_hip_events"), d.get("full_step_cfg4",{}).get("ms_per_step"), d.get("allreduce_us"))
    except Exception as e:
        print(f, "ERR", e)
PY
