#!/bin/bash
# counters of k_causal_conv<2,8> (24 -> 24 channels, PEMSD7 size): HBM-side bytes and L2 requests, separate passes
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
OUT=$O/causal_conv_pmc_raw.txt; : > $OUT
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  D=$O/pmc_tmp; rm -rf $D
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $D -- python3 $R/tools/causal_conv_time.py --reps 5 > /dev/null 2>&1
  echo "=== counters: $c" >> $OUT
  python3 $R/tools/pmc_kernel_table.py $(ls $D/*/*counter_collection.csv | head -1) k_causal_conv >> $OUT 2>&1
  rm -rf $D
done
cat $OUT
