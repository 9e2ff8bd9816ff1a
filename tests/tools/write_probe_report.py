import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
t = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "WRITE_SIZE":
        t[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
for k, v in t.items():
    print(k, [round(x / 1024, 2) for x in v], "MiB reported by WRITE_SIZE (KiB units); 256 MiB written")
