#!/bin/bash
# Manual tool: SQ counters of the inflate kernels alone (2 GiB of the mixed corpus, one wavefront to a chunk).
# usage (on the GPU box, from the repository root): bash tests/gpu_profile_inflate.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/inf_prof
mkdir -p $O
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $O/pmc_a -- python3 $R/tests/gpu_inflate_bench.py 2048 4 6 0 1 > /dev/null 2> $O/pmc_a.err || { tail -5 $O/pmc_a.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/pmc_b -- python3 $R/tests/gpu_inflate_bench.py 2048 4 6 0 1 > /dev/null 2> $O/pmc_b.err || { tail -5 $O/pmc_b.err; exit 1; }
cd $R
python3 profiles/summarize_pmc.py --counters $O/pmc_a $O/pmc_b > $O/sq_counters_inflate.txt || exit 1
rm -rf $O/pmc_a $O/pmc_b
