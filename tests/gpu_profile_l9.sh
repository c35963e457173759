# Kernel durations and SQ counters of the search along four-byte chains (level 9, low-entropy corpus).  Run through gpurun
# from the repository root; results land in gpurun_out/r03_l9/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_l9
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ZAMD_SERIALIZE=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --workload l9_lowent --chunks-per-gpu 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-serial-pass --no-inflate > $O/bench_line_under_rocprof.json 2> $O/prof.err || { tail -20 $O/prof.err; exit 1; }
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_serialized.csv \;
export ZAMD_SERIAL=1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_sqa -- python3 $R/bench.py --workload l9_lowent --steps 1 --warmup 0 --chunks-per-gpu 512 --no-cpu-baseline --no-inflate --no-serial-pass > /dev/null 2> $O/pmc_sqa.err || { tail -5 $O/pmc_sqa.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_sqb -- python3 $R/bench.py --workload l9_lowent --steps 1 --warmup 0 --chunks-per-gpu 512 --no-cpu-baseline --no-inflate --no-serial-pass > /dev/null 2> $O/pmc_sqb.err || { tail -5 $O/pmc_sqb.err; exit 1; }
unset ZAMD_SERIAL
cd $R
python3 profiles/summarize_pmc.py --counters $O/pmc_sqa $O/pmc_sqb > $O/sq_counters.txt || exit 1
rm -rf $O/prof $O/pmc_sqa $O/pmc_sqb
ls $O
