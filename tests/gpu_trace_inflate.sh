#!/bin/bash
# Kernel timeline of the inflate leg of the default workload.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --workload ${1:-l6_mixed} --chunks-per-gpu ${2:-8192} --steps 1 --warmup 1 --no-cpu-baseline --no-serial-pass > /tmp/tr.json 2>/tmp/tr.err || { tail -5 /tmp/tr.err; exit 1; }
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/trace_inflate.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r['Kernel_Name'].startswith('k_inf') or r['Kernel_Name'].startswith('void k_inf')]
t0 = min(int(r['Start_Timestamp']) for r in rows)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    print(f"{s:10.2f} {e:10.2f} {e-s:9.2f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:48]} grid={r.get('Grid_Size_X', r.get('Grid_Size','?'))}")
PY
tail -1 /tmp/tr.json | cut -c1-100
