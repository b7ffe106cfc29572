import csv, sys, collections
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_pairs"
rows = list(csv.DictReader(open(path)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
