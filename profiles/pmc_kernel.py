#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc pass:  python profiles/pmc_kernel.py <counter_collection.csv> <kernel-name substring>"""
import collections, csv, sys
tot, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
for (kn, c), v in sorted(tot.items()):
    print("%-60s %-24s launches %4d  avg %.4g" % (kn, c, cnt[(kn, c)], v / cnt[(kn, c)]))
