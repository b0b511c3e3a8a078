cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cp $R/ms_gat_amd/libmsgat_hip.so /tmp/orig.so
for v in orig nt; do echo "LIB=$v"; if [ $v = orig ]; then cp /tmp/orig.so $R/ms_gat_amd/libmsgat_hip.so; else cp $R/ms_gat_amd/libmsgat_lab_$v.so $R/ms_gat_amd/libmsgat_hip.so; fi
timeout -k 10 120 python3 $R/tools/kbench.py --only mix_,project_fwd --sets 3 2>&1 | grep "us "; done
cp /tmp/orig.so $R/ms_gat_amd/libmsgat_hip.so
