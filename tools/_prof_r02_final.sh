# end-of-round collection on the GPU box; outputs under gpurun_out/r02f (copied into profiles/r02 by hand)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench_plain.json 2> $O/bench_plain.err; echo "bench plain rc=$?"
MSGAT_BENCH_FORCE_DIST=1 python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_force_dist_1rank.json 2> $O/bench_force.err; echo "bench force rc=$?"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun.err; echo "bench torchrun rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-baselines --steps 20 --warmup 5 > $O/bench_hot_under_rocprof.json 2>/dev/null
KT=$(ls $O/bench_kt/*/*kernel_trace.csv | head -1)
python3 $R/tools/trace_one_step.py $O/bench_kt --per-step 24 --skip 89 > $O/hot_path_launches.txt
python3 $R/tools/roofline_trace_table.py $KT > $O/agg_lds_by_phase.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/hot_fetch -- python3 $R/bench.py --no-baselines --steps 5 --warmup 2 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/hot_write -- python3 $R/bench.py --no-baselines --steps 5 --warmup 2 > /dev/null 2>&1
python3 $R/tools/pmc_kernel_table.py $(ls $O/hot_fetch/*/*counter_collection.csv | head -1) > $O/pmc_fetch.txt
python3 $R/tools/pmc_kernel_table.py $(ls $O/hot_write/*/*counter_collection.csv | head -1) > $O/pmc_write.txt
for r in 3 5; do timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/full$r -- python3 $R/tools/full_model_profile.py --R $r --steps 6 > $O/full$r.log 2>&1; python3 $R/tools/trace_summary.py $O/full$r/*/*kernel_trace.csv k_adam_advance 1 3 70 > $O/full_step_R$r.txt; done
python3 $R/tools/full_model_profile.py --R 5 --steps 10 2>/dev/null | tail -1 > $O/full_step_unprofiled.txt
python3 $R/tools/full_model_profile.py --R 5 --steps 10 --graph 2>/dev/null | tail -1 >> $O/full_step_unprofiled.txt
python3 $R/tools/full_model_profile.py --R 3 --steps 10 2>/dev/null | tail -1 >> $O/full_step_unprofiled.txt
rm -rf $O/bench_kt/*/*agent* $O/hot_fetch $O/hot_write $O/full3 $O/full5
ls -la $O; cat $O/full_step_unprofiled.txt; tail -c 1500 $O/bench_plain.json
