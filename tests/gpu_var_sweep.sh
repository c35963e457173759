# A/B of library builds (zlib_amd/libzamd_<name>.so) on bench.py; writes gpurun_out/var_<name>.json
cp zlib_amd/libzamd.so /tmp/libzamd_base.so
for v in ${VS:-base vA vB}; do
  if [ $v = base ]; then cp /tmp/libzamd_base.so zlib_amd/libzamd.so; else cp zlib_amd/libzamd_$v.so zlib_amd/libzamd.so; fi
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || exit 1
done
