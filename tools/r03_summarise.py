#!/usr/bin/env python3
"""gpurun_out/r03/ (tools/r03_profiles.sh) -> the tracked summaries under profiles/r03_*."""
import csv
import glob
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(REPO, "gpurun_out", "r03"), os.path.join(REPO, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return f[0] if f else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:name.index("(")] if "(" in name and not name.startswith("(") else name


# kernel stats
for tag, out in (("bench_stats", "r03_bench_kernel_stats.csv"), ("stats_pointnet2", "r03_pointnet2_kernel_stats.csv"),
                 ("stats_dgcnn", "r03_dgcnn_kernel_stats.csv"), ("stats_gcnn", "r03_gcnn_kernel_stats.csv"),
                 ("stats_pointconv", "r03_pointconv_kernel_stats.csv")):
    f = one("%s/**/*kernel_stats.csv" % tag)
    if f:
        shutil.copy(f, os.path.join(DST, out))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, "r03_bench_under_rocprof.json"))
# counter summaries
summ = os.path.join(REPO, "tools", "pmc_summarise.py")
with open(os.path.join(DST, "r03_bench_pmc_summary.csv"), "w") as f:
    f.write(subprocess.run([sys.executable, summ, os.path.join(SRC, "bench_pmc")], capture_output=True, text=True).stdout)
with open(os.path.join(DST, "r03_models_pmc_summary.csv"), "w") as f:
    for m in ("pointnet2", "dgcnn", "gcnn", "pointconv"):
        f.write("# %s\n" % m)
        f.write(subprocess.run([sys.executable, summ, os.path.join(SRC, "pmc_" + m)], capture_output=True, text=True).stdout)
# PointNet++ sa1 gather A/B
rows = []
for v in ("reg", "walk"):
    per = {}
    for kind in ("fetch", "write", "tcc"):
        f = one("gather_%s_%s/**/*counter_collection.csv" % (kind, v))
        for r in csv.DictReader(open(f)):
            if "pt_gather_kernel" not in r["Kernel_Name"]:
                continue
            d = per.setdefault(int(r["Dispatch_Id"]), {})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
            d["us_" + kind] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    # dispatch ids differ between the three passes: align by order
    seqs = {}
    for kind, cols in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("tcc", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"])):
        seqs[kind] = [per[k] for k in sorted(per) if cols[0] in per[k]]
    n = min(len(s) for s in seqs.values())
    for i in range(n - 3, n):   # the three scales of the last step
        fd, wd, td = seqs["fetch"][i], seqs["write"][i], seqs["tcc"][i]
        rows.append([v, i - (n - 3) + 1, "%.1f" % fd["us_fetch"], "%.1f" % (fd["FETCH_SIZE"] / 1024), "%.1f" % (2 * fd["FETCH_SIZE"] / 1024),
                     "%.1f" % (wd["WRITE_SIZE"] / 1024), "%.3e" % td["TCC_REQ_sum"], "%.3f" % (td["TCC_HIT_sum"] / max(td["TCC_HIT_sum"] + td["TCC_MISS_sum"], 1))])
with open(os.path.join(DST, "r03_gather_ab.csv"), "w") as f:
    f.write("# tools/r03_profiles.sh: pt_gather_kernel of PointNet++ sa1 on the 3300-coalition Shapley step, region-reduced tables (reg) against the\n"
            "# member walk (walk, tuning key 5 = 21); separate rocprofv3 passes (--pmc FETCH_SIZE | WRITE_SIZE | TCC_*); FETCH_SIZE in KB x 1024,\n"
            "# raw and doubled (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section; includes Infinity-Cache hits)\n")
    f.write("variant,scale,us,fetch_MB_raw,fetch_MB_x2,write_MB,l2_requests,l2_hit_rate\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
with open(os.path.join(DST, "r03_knn_refine.txt"), "w") as f:
    f.write("# tools/r03_profiles.sh: DGCNN interaction step (12 000 coalitions): kNN counters (tuning key 4 = 3, two runs of the step), then the\n"
            "# step with the float32 ranking only (tuning key 5 = 20) and with the exact re-ranking of near-ties (default)\n")
    for name in ("knn_counters.log", "dgcnn_no_refine.log", "dgcnn_refine.log"):
        f.write(open(os.path.join(SRC, name)).read().strip().splitlines()[-1] + "\n")
print("written:", sorted(x for x in os.listdir(DST) if x.startswith("r03_")))
