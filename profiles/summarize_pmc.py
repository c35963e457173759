#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel.

usage: summarize_pmc.py INPUT_BYTES FETCH_DIR WRITE_DIR [--commit HASH] [--workload NAME] > hbm_traffic.json
FETCH_DIR / WRITE_DIR hold the output of two separate passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE).
FETCH_SIZE and WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts half the bytes of wide
coalesced reads, hence the factor 2 on the read side.
"""
import csv
import glob
import json
import sys


def collect(d, counter):
    acc = {}
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc.setdefault(name, [0.0, set()])
            a[0] += float(row["Counter_Value"])
            a[1].add(row.get("Dispatch_Id"))
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def all_counters(dirs):
    """--counters DIR...: every counter of every pass, summed over the dispatches of each kernel"""
    acc = {}
    for d in dirs:
        for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(fn)):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc.setdefault(name, {}).setdefault(row["Counter_Name"], 0.0)
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print("    %-26s %.4g" % (c, acc[k][c]))


def main():
    if sys.argv[1] == "--counters":
        return all_counters(sys.argv[2:])
    n = int(sys.argv[1])
    rd, wr = collect(sys.argv[2], "FETCH_SIZE"), collect(sys.argv[3], "WRITE_SIZE")
    extra = sys.argv[4:]
    opt = {extra[i].lstrip("-"): extra[i + 1] for i in range(0, len(extra) - 1, 2)}
    out = {"input_bytes": n, "commit": opt.get("commit", "unknown"), "workload": opt.get("workload", "l6_mixed"),
           "note": "FETCH_SIZE / WRITE_SIZE in KB from two separate --pmc passes; hbm_read_bytes = 2 * FETCH_SIZE * 1024 (gfx950)",
           "kernels": {}}
    for k in sorted(set(rd) | set(wr)):
        r = 2.0 * rd.get(k, (0, 0))[0] * 1024.0
        w = wr.get(k, (0, 0))[0] * 1024.0
        out["kernels"][k] = {"launches": max(rd.get(k, (0, 0))[1], wr.get(k, (0, 0))[1]), "hbm_read_bytes": r, "hbm_write_bytes": w,
                             "hbm_bytes_per_input_byte": round((r + w) / n, 3)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
