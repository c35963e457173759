# A/B of k_match variants (ZAMD_MV) on the default benchmark workload; writes gpurun_out/mv*.json
for mv in ${MVS:-0 4}; do
  ZAMD_MV=$mv timeout -k 10 200 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-inflate ${BENCH_ARGS:-} > gpurun_out/mv$mv.json 2> gpurun_out/mv$mv.err || exit 1
done
