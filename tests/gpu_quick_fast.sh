cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_deflate.py -x -q --timeout 300 --timeout-method thread > gpurun_out/q_tests.log 2>&1 || { tail -30 gpurun_out/q_tests.log; exit 1; }
tail -2 gpurun_out/q_tests.log
timeout -k 10 300 python3 bench.py --workload l1_text --steps 2 --warmup 1 --no-cpu-baseline --no-inflate --no-serial-pass > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err || { tail -5 gpurun_out/q_bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('gpurun_out/q_bench.json'));print(d['value'], d['ms_per_step'])"
