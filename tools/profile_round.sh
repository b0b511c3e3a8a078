#!/bin/bash
# Collects the profiles a round commits under profiles/rNN/.  Runs ON THE GPU BOX:
#     gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03 [parts]'
# parts (default "hot full stress pmc"; "pemsd4" on request):
#   pemsd4  kernel trace of `bench.py --workload pemsd4 --no-baselines` (configs[1]): per-kernel totals per step and one
#           step in launch order with the gaps between launches (the eager step is host-bound there)
#   hot     rocprofv3 kernel trace of `bench.py --no-baselines` (the hot-path step): per-kernel totals per step,
#           one step in launch order, the attention-aggregate kernel by phase (cold loop / in step / cached loop)
#   full    kernel traces of whole training steps (tools/full_model_profile.py, R = 3 and R = 5) + unprofiled wall times
#   stress  kernel trace of the N = 8192 stress kernels (tools/stress_kernels.py)
#   pmc     FETCH_SIZE / WRITE_SIZE in SEPARATE passes (kernel trace only: never with --sys-trace etc.), for the
#           hot-path kernels (tools/kbench.py --eager) and the stress kernels
# Everything lands in gpurun_out/<round>/ (merged back by gpurun); copy what is to be judged into profiles/<round>/.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; RND=${1:-r03}; PARTS=${2:-"hot full stress pmc"}; O=$R/gpurun_out/$RND; mkdir -p $O
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has hot; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-baselines --steps 20 --warmup 5 > $O/bench_hot_under_rocprof.json 2>/dev/null
  KT=$(ls $O/bench_kt/*/*kernel_trace.csv | head -1)
  # (anchor: a kernel launched once per step -- k_project_mfma, the second depth's projection; k_qonly and k_agg_proj are
  # gone from this workload since round 4: the first depth's forward is one launch)
  python3 $R/tools/trace_summary.py $KT k_project_mfma 1 15 40 > $O/hot_path_per_step.txt
  python3 $R/tools/trace_one_step.py $O/bench_kt --anchor k_project_mfma > $O/hot_path_launches.txt 2>&1
  python3 $R/tools/roofline_trace_table.py $KT > $O/aggregate_by_phase.txt
  cp $(ls $O/bench_kt/*/*kernel_stats.csv | head -1) $O/hot_path_kernel_stats.csv 2>/dev/null
  rm -rf $O/bench_kt
fi
if has pemsd4; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p4_kt -- python3 $R/bench.py --workload pemsd4 --no-baselines --steps 50 --warmup 10 > $O/pemsd4_bench_under_rocprof.json 2>/dev/null
  KT=$(ls $O/p4_kt/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_summary.py $KT k_project_mfma 1 30 40 > $O/pemsd4_per_step.txt
  python3 $R/tools/trace_one_step.py $O/p4_kt --anchor k_project_mfma > $O/pemsd4_launches.txt 2>&1
  cp $(ls $O/p4_kt/*/*kernel_stats.csv | head -1) $O/pemsd4_kernel_stats.csv 2>/dev/null
  rm -rf $O/p4_kt
fi
if has full; then
  for r in 3 5; do
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/full$r -- python3 $R/tools/full_model_profile.py --R $r --steps 6 > $O/full$r.log 2>&1
    python3 $R/tools/trace_summary.py $O/full$r/*/*kernel_trace.csv k_adam_advance 1 3 70 > $O/full_step_R${r}_kernels.txt
    rm -rf $O/full$r
  done
  for args in "--R 5" "--R 5 --graph" "--R 3" "--R 3 --graph"; do python3 $R/tools/full_model_profile.py $args --steps 10 2>/dev/null | tail -1; done > $O/full_step_wall.txt
fi
if has stress; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stress_kt -- python3 $R/tools/stress_kernels.py --reps 3 > $O/stress_kernels.log 2>&1
  cp $(ls $O/stress_kt/*/*kernel_stats.csv | head -1) $O/stress_kernel_stats.csv 2>/dev/null
  rm -rf $O/stress_kt
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/hot_$c -- python3 $R/tools/kbench.py --eager --reps 5 > /dev/null 2>&1
    python3 $R/tools/pmc_kernel_table.py $(ls $O/hot_$c/*/*counter_collection.csv | head -1) > $O/pmc_hot_$c.txt
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/stress_$c -- python3 $R/tools/stress_kernels.py --reps 2 > /dev/null 2>&1
    python3 $R/tools/pmc_kernel_table.py $(ls $O/stress_$c/*/*counter_collection.csv | head -1) > $O/pmc_stress_$c.txt
    rm -rf $O/hot_$c $O/stress_$c
  done
fi
ls -la $O
