import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            if "inflate" in k: print("%-28s %-22s n=%d avg=%.4g max=%.4g" % (k[:28], c, len(v), sum(v)/len(v), max(v)))
