#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection.csv files: mean counter value per kernel."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "nvllm" not in k:
                continue
            short = k.split("(")[0].split("::")[-1][:40] if "(" in k else k[:60]
            agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
            dur[short].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (k, c), v in sorted(agg.items()):
            print("%-44s %-28s n=%-4d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
        for k, v in dur.items():
            print("%-44s duration_ns mean=%.0f" % (k, sum(v) / len(v)))
