#!/usr/bin/env python3
"""Per-kernel totals of the TIMED steps of a `rocprofv3 --kernel-trace` run of bench.py.

    python profiles/summarize_trace.py <..._kernel_trace.csv> --marker stem_fwd --per-step 2 --warmup 2 --steps 5

The trace is cut at the first launch of `marker` that belongs to the first timed step (bench.py's warm-up
steps hold MIOpen's one-off solver search for the stock-torch discriminators); `--per-step` = launches of the
marker kernel per step (prototype_full: 2 grad-mode generator forwards; source_only: 1)."""
import argparse
import collections
import csv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="stem_fwd")
    ap.add_argument("--per-step", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--by-grid", default="", help="comma-separated kernel-name substrings: also list these kernels per launch grid")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else str(r.get("Grid_Size", "?"))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    first = marks[a.warmup * a.per_step]
    rows = rows[first:]
    tot, cnt = collections.Counter(), collections.Counter()
    gt, gc = collections.Counter(), collections.Counter()
    subs = [x for x in a.by_grid.split(",") if x]
    for s, e, n, g in rows:
        tot[n] += e - s
        cnt[n] += 1
        if any(x in n for x in subs):
            gt[(n, g)] += e - s
            gc[(n, g)] += 1
    total = sum(tot.values())
    print("kernel time %.1f ms/step, %d dispatches/step" % (total / 1e6 / a.steps, len(rows) // a.steps))
    print("%-90s %6s %9s %9s %6s" % ("kernel", "calls", "ms/step", "avg_us", "%"))
    for n, t in tot.most_common(a.top):
        print("%-90s %6d %9.2f %9.1f %6.1f" % (n[:90], cnt[n], t / 1e6 / a.steps, t / 1e3 / cnt[n], 100.0 * t / total))
    if subs:
        print("\nper launch grid (threads x, y, z):")
        for (n, g), t in gt.most_common():
            print("%-60s %-18s %6d %9.2f %9.1f" % (n[:60], g, gc[(n, g)], t / 1e6 / a.steps, t / 1e3 / gc[(n, g)]))


if __name__ == "__main__":
    main()

