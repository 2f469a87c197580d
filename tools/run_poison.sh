#!/bin/bash
# The small-batch pipelines (one core and several cores on one GPU) with every device allocation poisoned: results must not change.
cd "$(dirname "$0")/.."
for pz in ${POISONS:-0xff 0xa5 0x01 0x80}; do
  echo "== BASAL_POISON=$pz"
  BASAL_POISON=$pz timeout -k 10 ${TMO:-400} python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_pipe.py -q -x -m gpu -p no:cacheprovider -k "${KEXPR:-two_ranks or pipe}" 2>&1 | tail -15
done
