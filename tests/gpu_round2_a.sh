# first GPU pass of the round: the tests that changed, a default bench line, and the serial-stream kernel trace
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_bench_contract.py -x -q --timeout 300 --timeout-method thread > gpurun_out/a_tests.log 2>&1 || { tail -30 gpurun_out/a_tests.log; exit 1; }
tail -3 gpurun_out/a_tests.log
timeout -k 10 500 python3 bench.py > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err || { tail -20 gpurun_out/a_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_serial
ZAMD_SERIALIZE=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_serial -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass > $R/gpurun_out/a_serial_line.json 2> $R/gpurun_out/a_serial.err || { tail -20 $R/gpurun_out/a_serial.err; exit 1; }
find $R/gpurun_out/prof_serial -name '*kernel_stats.csv' -exec cp {} $R/gpurun_out/a_kernel_stats_serial.csv \;
head -c 1500 $R/gpurun_out/a_bench.json
