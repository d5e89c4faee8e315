#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel.  usage: pmc_summarise.py <dir> [<dir> ...] [--min-ms 0.5]"""
import csv
import glob
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
dur = defaultdict(float)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = k[:k.index("(")] if "(" in k and not k.startswith("(") else k
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"):
                    pass
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                calls[k] += 1
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
names = sorted({c for v in agg.values() for c in v})
print("kernel,dispatches,ms," + ",".join(names) + ",mfma_busy_frac,clock_GHz")
for k in sorted(agg, key=lambda k: -dur[k]):
    v = agg[k]
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024) if gui else 0.0
    ghz = gui / 8 / (dur[k] * 1e-3) / 1e9 if dur[k] else 0.0
    print('"%s",%d,%.3f,' % (k, calls[k], dur[k]) + ",".join("%.4e" % v.get(c, 0.0) for c in names) + ",%.3f,%.3f" % (busy, ghz))
