#!/usr/bin/env python3
"""rocprofv3 counter_collection.csv -> mean counter value per (kernel, grid size), KiB units as reported."""
import collections, csv, glob, re, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            agg[(k, int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, g, c), v in sorted(agg.items()):
            print("%-40s grid=%-10d %-12s n=%-3d mean=%.6g" % (k, g, c, len(v), sum(v) / len(v)))
