#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters over several passes (each pass = one directory with its own counters).
usage: pmc_per_dispatch.py <dir> [<dir> ...] [--min-ms 0.3]   -> CSV: kernel, dispatches, mean ms, mean of every counter per dispatch
(FETCH_SIZE / WRITE_SIZE are in KB: printed as bytes; `fetch_bytes_x2` applies the gfx950 correction for wide coalesced reads,
MI355X_MICROARCH.md: the counter tallies 128-B requests at 64 B)."""
import csv
import glob
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
min_ms = float(sys.argv[sys.argv.index("--min-ms") + 1]) if "--min-ms" in sys.argv else 0.3
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = k[:k.index("(")] if "(" in k and not k.startswith("(") else k
            c = r["Counter_Name"]
            tot[k][c] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if (key, c) not in seen:
                seen.add((key, c))
                cnt[k][c] += 1
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
names = sorted({c for v in tot.values() for c in v})
extra = ["fetch_bytes_x2", "write_bytes", "l2_hit_rate"]
print("kernel,dispatches_per_pass,mean_ms," + ",".join(names) + "," + ",".join(extra))
for k in sorted(tot, key=lambda k: -sum(dur[k])):
    ms = sum(dur[k]) / max(len(dur[k]), 1)
    if ms < min_ms:
        continue
    mean = {c: tot[k][c] / max(cnt[k][c], 1) for c in names}
    n = max(cnt[k].values())
    fx = 2048.0 * mean.get("FETCH_SIZE", 0.0) if "FETCH_SIZE" in tot[k] else float("nan")
    wb = 1024.0 * mean.get("WRITE_SIZE", 0.0) if "WRITE_SIZE" in tot[k] else float("nan")
    h, m = mean.get("TCC_HIT_sum", 0.0), mean.get("TCC_MISS_sum", 0.0)
    hr = h / (h + m) if h + m > 0 else float("nan")
    print('"%s",%d,%.3f,' % (k, n, ms) + ",".join("%.4e" % mean.get(c, float("nan")) for c in names) + ",%.4e,%.4e,%.3f" % (fx, wb, hr))
