import csv, sys, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "*/*counter_collection.csv") + glob.glob(sys.argv[1] + "*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-48:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "gconv_kernel" in k:
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-34s %14.5g  (x%d)" % (c, sum(v) / len(v), len(v)))
