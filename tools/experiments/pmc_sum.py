import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:34]
            if k.startswith(("k_", "void k_")):
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            print(d, k, {a: "%.4g" % b for a, b in v.items()})
