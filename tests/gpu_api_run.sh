cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_zlib_api.py tests/test_gpu_gzfile.py tests/test_gpu_minizip.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread > gpurun_out/api_tests.log 2>&1 || { tail -30 gpurun_out/api_tests.log; exit 1; }
tail -2 gpurun_out/api_tests.log
ZAMD_Z_TRACE=1 timeout -k 10 300 python3 -u tests/gpu_api_bench.py 2048 6 2> gpurun_out/api_trace.log || { tail gpurun_out/api_trace.log; exit 1; }
grep pipelined gpurun_out/api_trace.log | tail -2
for b in 256 1024; do echo batch $b; ZAMD_Z_BATCH_MIB=$b timeout -k 10 300 python3 -u tests/gpu_api_bench.py 3584 6 | grep "deflate()" || exit 1; done
