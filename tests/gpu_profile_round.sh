# Collects what profiles/ holds: kernel stats of the default bench run, the bench lines, HBM traffic (two --pmc
# passes) and the SQ counters of k_match (two --pmc passes).  Run through gpurun from the repository root.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_*
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_line.json 2> $R/gpurun_out/prof.err || exit 1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 2048 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_fetch.err || exit 1
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 2048 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_write.err || exit 1
export ZAMD_SERIAL=1
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sqa -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 512 --no-cpu-baseline --no-inflate > /dev/null 2> $R/gpurun_out/pmc_sqa.err || exit 1
timeout -s KILL 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_sqb -- python3 $R/bench.py --steps 1 --warmup 0 --chunks-per-gpu 512 --no-cpu-baseline --no-inflate > /dev/null 2> $R/gpurun_out/pmc_sqb.err || exit 1
unset ZAMD_SERIAL
cd $R
python3 profiles/summarize_pmc.py $((2048 << 20)) gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/hbm_traffic.json || exit 1
python3 profiles/summarize_pmc.py --counters gpurun_out/pmc_sqa gpurun_out/pmc_sqb > gpurun_out/sq_counters.txt || exit 1
timeout -k 10 500 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
timeout -k 10 500 python3 bench.py --workload l1_text --chunks-per-gpu 4096 > gpurun_out/bench_l1.json 2> gpurun_out/bench_l1.err || exit 1
timeout -k 10 500 python3 bench.py --workload l9_lowent --chunks-per-gpu 4096 --no-cpu-baseline > gpurun_out/bench_l9.json 2> gpurun_out/bench_l9.err || exit 1
find gpurun_out/prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/kernel_stats.csv \;
ls gpurun_out/prof/*/
