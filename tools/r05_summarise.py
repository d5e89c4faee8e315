#!/usr/bin/env python3
"""gpurun_out/r05f/ (tools/r05_profiles.sh) -> the tracked summaries under profiles/r05_*."""
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(REPO, "gpurun_out", "r05f"), os.path.join(REPO, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(SRC, pattern), recursive=True)       # (gpurun merges every run into the same tree: take the newest)
    return max(f, key=os.path.getmtime) if f else None


for tag, out in (("bench_stats", "r05_bench_kernel_stats.csv"), ("stats_pointnet2", "r05_pointnet2_kernel_stats.csv"),
                 ("stats_dgcnn", "r05_dgcnn_kernel_stats.csv"), ("stats_gcnn", "r05_gcnn_kernel_stats.csv"),
                 ("stats_pointconv", "r05_pointconv_kernel_stats.csv")):
    f = one("%s/**/*kernel_stats.csv" % tag)
    if f:
        shutil.copy(f, os.path.join(DST, out))
for name, out in (("bench_under_rocprof.json", "r05_bench_under_rocprof.json"), ("bench.json", "r05_bench.json"),
                  ("bench_2rank_rehearsal.json", "r05_bench_2rank_rehearsal.json"), ("bench_sweep_1gpu.json", "r05_bench_sweep_1gpu.json"),
                  ("bench_sweep_2rank_rehearsal.json", "r05_bench_sweep_2rank_rehearsal.json")):
    src = os.path.join(SRC, name)
    if os.path.exists(src):
        lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
        if lines:
            with open(os.path.join(DST, out), "w") as f:
                f.write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
summ = os.path.join(REPO, "tools", "pmc_summarise.py")
with open(os.path.join(DST, "r05_bench_pmc_summary.csv"), "w") as f:
    f.write(subprocess.run([sys.executable, summ, os.path.join(SRC, "bench_pmc")], capture_output=True, text=True).stdout)
with open(os.path.join(DST, "r05_models_pmc_summary.csv"), "w") as f:
    for m in ("pointnet2", "dgcnn", "gcnn", "pointconv"):
        f.write("# %s\n" % m)
        f.write(subprocess.run([sys.executable, summ, os.path.join(SRC, "pmc_" + m)], capture_output=True, text=True).stdout)
print("written:", sorted(x for x in os.listdir(DST) if x.startswith("r05_")))
