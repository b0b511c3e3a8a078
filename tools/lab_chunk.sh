#!/bin/bash
# Lab run (GPU box): the one-pass 1x1-convolution backward (k_chanpair_glds, contraction + mix output) with FEW mix
# channels per block -- nzb z-blocks over B per run, each staging all of A (second and later readers from the XCD's L2)
# and writing its 12-37 dx rows over the run's whole position range -- against the round-3/4 form (one block = all 72
# rows).  Needs `python -m ms_gat_amd.build --lab` (build/lab/libmsgat_lab.so, -DMSGAT_LAB: MSGAT_LAB_NZB = z-blocks
# over B, MSGAT_LAB_BPC = resident blocks per CU).
#     gpurun --timeout 1100 -- 'bash tools/lab_chunk.sh r05'
set -u
R=$GRAFT_REPO_ROOT; RND=${1:-r05}; O=$R/gpurun_out/$RND; mkdir -p $O
LAB=$R/build/lab/libmsgat_lab.so
OUT=$O/chunk_lab_raw.txt; : > $OUT
cd $R
for cfg in "0 1" "5 1" "5 2" "3 2" "2 2"; do
  set -- $cfg
  export MSGAT_LAB_NZB=$1 MSGAT_LAB_BPC=$2
  echo "=== nzb=$1 blocks_per_cu=$2" >> $OUT
  timeout -k 10 200 python3 tools/kbench.py --lib $LAB --sets 4 --only project_bwd,cmix98,cmix72 >> $OUT 2>&1 || { echo "kbench failed" >> $OUT; exit 1; }
  MSGAT_TEST_LIB=$LAB timeout -k 10 300 python3 -m pytest tests/test_gpu_branches.py -q -x -m gpu -k "contract_mix_segments or stage_project_backward" 2>&1 | tail -3 >> $OUT || { echo "tests failed" >> $OUT; exit 1; }
done
for cfg in "0 1" "5 2" "3 2" "2 2"; do
  set -- $cfg
  export MSGAT_LAB_NZB=$1 MSGAT_LAB_BPC=$2
  echo "=== in the steps: nzb=$1 blocks_per_cu=$2" >> $OUT
  timeout -k 10 200 python3 tools/hot_kernels.py --lib $LAB 2>&1 | grep -E "chanpair|aggfirst|busy" >> $OUT
  timeout -k 10 200 python3 tools/full_model_profile.py --lib $LAB --R 3 --steps 10 2>&1 | tail -1 >> $OUT
done
cat $OUT
