cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_fast
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_fast -o fast --output-format csv -- python3 $R/bench.py --workload l1_text --chunks-per-gpu 1024 --steps 1 --warmup 0 --no-cpu-baseline --no-inflate --no-serial-pass > $R/gpurun_out/pmc_fast/run.log 2>&1 || { tail -5 $R/gpurun_out/pmc_fast/run.log; exit 1; }
ls -R $R/gpurun_out/pmc_fast | head -20
