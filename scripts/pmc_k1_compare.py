#!/usr/bin/env python3
"""K1 lab under rocprofv3 --pmc: per variant (dispatches of flash_attn_mfma_kernel in launch order,
`per` launches each) the mean of every counter, plus matrix-pipe busy / co-issue fractions.
usage: pmc_k1_compare.py per name1,name2,... dir1 [dir2 ...]"""
import collections, csv, glob, sys
per, names, dirs = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
tot = {n: collections.OrderedDict() for n in names}
for d in dirs:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "flash_attn_mfma_kernel" in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        rank = {i: k for k, i in enumerate(ids)}
        agg = collections.defaultdict(list)
        for r in rows:
            v = rank[int(r["Dispatch_Id"])] // per
            if v < len(names):
                agg[(names[v], r["Counter_Name"])].append(float(r["Counter_Value"]))
                agg[(names[v], "duration_ns:" + d.rsplit("/", 1)[-1])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (n, c), v in agg.items():
            tot[n][c] = sum(v) / len(v)
for n in names:
    t = tot[n]
    print("== %s" % n)
    for c, v in t.items():
        print("   %-34s %.5g" % (c, v))
    if "GRBM_GUI_ACTIVE" in t and "SQ_VALU_MFMA_BUSY_CYCLES" in t:
        cyc = t["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        simd_cycles = cyc * 1024          # 256 CUs x 4 SIMDs
        print("   cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs)      %.4g" % cyc)
        print("   matrix pipe busy  = MFMA_BUSY / (1024 SIMDs x cycles) = %.3f" % (t["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles))
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in t:
            print("   VALU co-issued under an MFMA = COEXEC / MFMA_BUSY         = %.3f" % (t["SQ_VALU_MFMA_COEXEC_CYCLES"] / t["SQ_VALU_MFMA_BUSY_CYCLES"]))
