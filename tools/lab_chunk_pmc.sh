#!/bin/bash
# Counter passes for tools/lab_chunk.sh's variants (GPU box): HBM-side bytes and L2 hit rate of the one-pass convolution
# backward with nzb z-blocks over B.  FETCH_SIZE / WRITE_SIZE / TCC hit+miss in SEPARATE passes, kernel trace only.
#     gpurun --timeout 900 -- 'bash tools/lab_chunk_pmc.sh r05'
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; RND=${1:-r05}; O=$R/gpurun_out/$RND; mkdir -p $O
LAB=$R/build/lab/libmsgat_lab.so
OUT=$O/chunk_lab_pmc_raw.txt; : > $OUT
for cfg in "0 1" "5 2" "2 2"; do
  set -- $cfg
  export MSGAT_LAB_NZB=$1 MSGAT_LAB_BPC=$2
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    D=$O/pmc_tmp; rm -rf $D
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $D -- python3 $R/tools/kbench.py --eager --reps 5 --sets 4 --lib $LAB --only project_bwd,cmix98 > /dev/null 2>&1
    echo "=== nzb=$1 blocks_per_cu=$2 counters: $c" >> $OUT
    python3 $R/tools/pmc_kernel_table.py $(ls $D/*/*counter_collection.csv | head -1) k_chanpair_glds >> $OUT 2>&1
    rm -rf $D
  done
done
cat $OUT
