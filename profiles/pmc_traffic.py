#!/usr/bin/env python3
"""HBM traffic per launch of the dominant kernel from two rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python profiles/pmc_traffic.py out/f/..._counter_collection.csv out/w/..._counter_collection.csv --match 'igemm_conv_ws_kernel<3'
(--match takes comma-separated substrings; --calls N divides the total by the number of conv calls instead of kernel launches)

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of a wide coalesced read
at 64 B, so the read side is doubled (the operand loads of this kernel are 16 B per lane)."""
import argparse
import collections
import csv
import json


def per_kernel(path, counter, match):
    tot, cnt = collections.Counter(), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter or not any(m in r["Kernel_Name"] for m in match.split(",")):
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--match", default="igemm_conv_ws_kernel<3")
    ap.add_argument("--out", default=None)
    ap.add_argument("--calls", type=int, default=0,
                    help="conv CALLS of the profiled run (bench.py's roofline.launches per step x steps incl. warm-up): a call of the bf16x3 "
                         "kernel may be a main launch + a K-split tail launch + its reduce; bytes are then reported per call")
    a = ap.parse_args()
    ft, fc = per_kernel(a.fetch_csv, "FETCH_SIZE", a.match)
    wt, wc = per_kernel(a.write_csv, "WRITE_SIZE", a.match)
    rows = {}
    for k in sorted(ft):
        rows[k] = {"launches": fc[k], "fetch_kib_raw_per_launch": ft[k] / fc[k],
                   "write_kib_per_launch": wt[k] / max(wc[k], 1),
                   "hbm_bytes_per_launch": (2.0 * ft[k] / fc[k] + wt[k] / max(wc[k], 1)) * 1024.0}
    n = sum(fc.values())
    total = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows.values())
    calls = a.calls if a.calls > 0 else n
    res = {"match": a.match, "launches": n, "calls": calls, "hbm_bytes_per_launch": total / max(calls, 1),
           "correction": "2 x FETCH_SIZE (gfx950 wide-read tally) + WRITE_SIZE, KiB -> bytes", "kernels": rows}
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
