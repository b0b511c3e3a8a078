#!/bin/bash
# Repeats the tests of the split-operand dense passes (LDS-DMA double buffer issued from inline asm: the kernels order buffer
# use themselves) in fresh processes, one after the other; a miscounted wait would show as a run-to-run difference.
#     gpurun --timeout 900 -- 'bash tools/soak_dense.sh 3'
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-3}); do
  timeout -k 10 280 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "stress or split or payload or reproducible or saturated or large_scores" 2>&1 | tail -1 || exit 1
done
