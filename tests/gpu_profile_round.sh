R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_line.json 2> $R/gpurun_out/prof.err || exit 1
cd $R
timeout -k 10 500 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
timeout -k 10 500 python3 bench.py --workload l1_text --chunks-per-gpu 4096 > gpurun_out/bench_l1.json 2> gpurun_out/bench_l1.err || exit 1
timeout -k 10 500 python3 bench.py --workload l9_lowent --chunks-per-gpu 4096 --no-cpu-baseline > gpurun_out/bench_l9.json 2> gpurun_out/bench_l9.err || exit 1
ls gpurun_out/prof/*/
