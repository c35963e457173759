# Collects what profiles/r03_* holds.  Run through gpurun from the repository root with the commit hash as the argument
# (the copy on the GPU box has no .git); results land in gpurun_out/r03/.
COMMIT=${1:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. every kernel alone on the GPU (the same batches issued to one stream): the durations the roofline is priced with
ZAMD_SERIALIZE=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass > $O/bench_line_serialized_under_rocprof.json 2> $O/prof_serial.err || { tail -20 $O/prof_serial.err; exit 1; }
find $O/prof_serial -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_serialized.csv \;
# 2. the default run under the profiler (four streams, kernels overlap)
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_overlapped.csv \;
# 3. HBM traffic: two separate counter passes
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 2048 --no-cpu-baseline --no-serial-pass > /dev/null 2> $O/pmc_fetch.err || { tail -5 $O/pmc_fetch.err; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 2048 --no-cpu-baseline --no-serial-pass > /dev/null 2> $O/pmc_write.err || { tail -5 $O/pmc_write.err; exit 1; }
# 4. SQ counters, one stream
export ZAMD_SERIAL=1
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_sqa -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 512 --no-cpu-baseline --no-inflate --no-serial-pass > /dev/null 2> $O/pmc_sqa.err || { tail -5 $O/pmc_sqa.err; exit 1; }
timeout -s KILL 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_sqb -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 512 --no-cpu-baseline --no-inflate --no-serial-pass > /dev/null 2> $O/pmc_sqb.err || { tail -5 $O/pmc_sqb.err; exit 1; }
unset ZAMD_SERIAL
cd $R
python3 profiles/summarize_pmc.py $((2048 << 20)) $O/pmc_fetch $O/pmc_write --commit $COMMIT --workload l6_mixed > $O/hbm_traffic.json || exit 1
python3 profiles/summarize_pmc.py --counters $O/pmc_sqa $O/pmc_sqb > $O/sq_counters.txt || exit 1
# 5. the bench lines
timeout -k 10 500 python3 bench.py > $O/bench_line.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
timeout -k 10 500 python3 bench.py --cpu-sample-mib 64 --workload l1_text > $O/bench_l1_text.json 2> $O/bench_l1.err || { tail -5 $O/bench_l1.err; exit 1; }
timeout -k 10 500 python3 bench.py --cpu-sample-mib 64 --workload l9_lowent --chunks-per-gpu 4096 > $O/bench_l9_lowent.json 2> $O/bench_l9.err || { tail -5 $O/bench_l9.err; exit 1; }
timeout -k 10 500 python3 bench.py --cpu-sample-mib 64 --workload l6_text > $O/bench_l6_text.json 2> $O/bench_l6t.err || { tail -5 $O/bench_l6t.err; exit 1; }
timeout -k 10 500 python3 bench.py --cpu-sample-mib 64 --workload l1_text --chunks-per-gpu 4096 > $O/bench_l1_text_4gib.json 2> $O/bench_l1b.err || { tail -5 $O/bench_l1b.err; exit 1; }
for w in l9_image_filtered l6_lowent_huff l6_image_rle; do
  timeout -k 10 500 python3 bench.py --cpu-sample-mib 64 --workload $w --chunks-per-gpu 4096 > $O/bench_$w.json 2> $O/bench_w.err || { tail -5 $O/bench_w.err; exit 1; }
done
for w in l2_text l3_text; do
  timeout -k 10 500 python3 bench.py --cpu-sample-mib 64 --workload $w > $O/bench_$w.json 2> $O/bench_w.err || { tail -5 $O/bench_w.err; exit 1; }
done
# 6. the stream API on host buffers
timeout -k 10 300 python3 -u tests/gpu_api_bench.py 3584 6 > $O/api_host_buffers.txt 2>&1 || { tail -5 $O/api_host_buffers.txt; exit 1; }
timeout -k 10 300 python3 -u tests/gpu_api_bench.py 3584 1 > $O/api_host_buffers_l1.txt 2>&1 || { tail -5 $O/api_host_buffers_l1.txt; exit 1; }
export PYTHONPATH=$R
{ timeout -k 10 200 python3 -u tests/gpu_threads_bench.py 4 6 && timeout -k 10 200 python3 -u tests/gpu_threads_bench.py 64 6; } > $O/api_threads.txt 2>&1 || { tail -5 $O/api_threads.txt; exit 1; }
# 7. batch inflate with several wavefronts to a chunk
{ timeout -k 10 300 python3 -u tests/gpu_inflate_bench.py 8192 4 6 0 1 2 3 4 && timeout -k 10 200 python3 -u tests/gpu_inflate_bench.py 512 4 6 0 1 4 8 16 && timeout -k 10 200 python3 -u tests/gpu_inflate_bench.py 2048 3 6 2 1 4; } > $O/inflate_parts.txt 2>&1 || { tail -5 $O/inflate_parts.txt; exit 1; }
rm -rf $O/prof_serial $O/prof $O/pmc_fetch $O/pmc_write $O/pmc_sqa $O/pmc_sqb
ls -la $O
