#!/bin/bash
# Manual tool: batch inflate with and without literal pairs on the corpora of the benchmark's workloads.
set -e
for P in 2; do
  export ZAMD_INF_PAIRS=$P
  echo "=== PAIRS $P"
  for args in "2048 4 6 0" "2048 0 6 0" "2048 0 1 0" "2048 2 6 2" "2048 2 9 0" "2048 3 6 3" "2048 3 9 1"; do
    echo "--- MiB kind level strategy: $args"
    timeout -k 10 200 python -u tests/gpu_inflate_bench.py $args 8 2>&1 | grep -E "inflate|round trip" | awk 'NR%4==3 || /round/'
  done
done
