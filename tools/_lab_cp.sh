cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/step; mkdir -p $O
cp $R/ms_gat_amd/libmsgat_hip.so /tmp/orig.so
for v in orig 896 768 640 512; do echo "LIB=$v"; if [ $v = orig ]; then cp /tmp/orig.so $R/ms_gat_amd/libmsgat_hip.so; else cp $R/ms_gat_amd/libmsgat_lab_$v.so $R/ms_gat_amd/libmsgat_hip.so; fi
rm -rf $O/kt_$v; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -- python3 $R/bench.py --no-baselines --steps 20 --warmup 5 > $O/bench_$v.json 2>/dev/null
python3 $R/tools/trace_one_step.py $O/kt_$v --per-step 29 --skip 89 | grep "agg_sddmm\|busy"; done
cp /tmp/orig.so $R/ms_gat_amd/libmsgat_hip.so
