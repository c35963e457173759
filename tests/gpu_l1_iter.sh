#!/bin/bash
# Level-1 iteration on the GPU box: parity tests of the deflate path, then the level-1 text workload at two sizes.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_deflate.py -x -q --timeout 180 --timeout-method thread > gpurun_out/l1_tests.log 2>&1
rc=$?; tail -3 gpurun_out/l1_tests.log
[ $rc -ne 0 ] && exit $rc
for c in 8192 4096; do
  timeout -k 10 300 python -u bench.py --workload l1_text --chunks-per-gpu $c --steps 3 --warmup 1 --no-cpu-baseline --no-serial-pass 2> gpurun_out/l1_bench_$c.err | tee gpurun_out/l1_bench_$c.json || exit 1
done
