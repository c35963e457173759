for mv in 0 1 2 3 4; do
  ZAMD_MV=$mv ZAMD_SERIAL=1 timeout -k 10 120 python bench.py --steps 2 --warmup 1 --chunks-per-gpu 2048 --no-cpu-baseline --no-inflate > gpurun_out/mv$mv.json 2> gpurun_out/mv$mv.err || exit 1
done
