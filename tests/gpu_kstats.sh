#!/bin/bash
# Per-kernel durations of the default workload with every kernel alone on the GPU (one stream).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks && ZAMD_SERIALIZE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${1:-l6_mixed} --steps 1 --warmup 1 --no-cpu-baseline --no-serial-pass --no-inflate > /tmp/ks.json 2>/tmp/ks.err || { tail -5 /tmp/ks.err; exit 1; }
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 - "$f" <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/kstats.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith('k_'):
        print(r['Name'][:40].ljust(40), r['Calls'].rjust(4), round(int(r['TotalDurationNs'])/1e6/ (int(r['Calls'])/4),1), 'ms per step', round(float(r['AverageNs'])/1e6,2))
PY
