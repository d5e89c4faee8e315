#!/usr/bin/env python3
"""gpurun_out/pmc_stall_{dgcnn3,gcnn3}/{a,b} (bash tools/pmc_stall.sh dgcnn3 --model dgcnn --mode interaction; ... gcnn3 ...)
-> profiles/r03_dgcnn_instruction_mix.csv: VALU instructions per MFMA and LDS conflict cycles per LDS instruction of the DGCNN /
GCNN kernels (one 12 000-coalition interaction step; two rocprofv3 --pmc passes)."""
import csv
import io
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = ["# tools/pmc_stall.sh + tools/r03_instruction_mix.py: one 12 000-coalition interaction step under two rocprofv3 --pmc passes.",
         "# SQ_INSTS_VALU counts MFMAs too; 'valu_per_mfma' = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA.  On gfx950 a VALU instruction",
         "# next to an fp32 MFMA stream costs ~5-9 cycles, an MFMA 64: at 15 VALU per MFMA (knn_kernel<64>) the VALU work is larger than the MFMA work.",
         "model,kernel,dispatches,ms,SQ_INSTS_MFMA,SQ_INSTS_VALU,valu_per_mfma,SQ_INSTS_LDS,lds_conflict_cycles_per_lds_inst,mfma_busy_frac"]
for model in ("dgcnn", "gcnn"):
    d = os.path.join(REPO, "gpurun_out", "pmc_stall_%s3" % model)
    txt = subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_summarise.py"), d + "/a", d + "/b"], capture_output=True, text=True).stdout
    for r in list(csv.DictReader(io.StringIO(txt)))[:6]:
        mf, va, lds, bc = (float(r[k]) for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"))
        lines.append('%s,"%s",%s,%s,%.4e,%.4e,%.2f,%.4e,%.2f,%s' % (model, r["kernel"], r["dispatches"], r["ms"], mf, va,
                                                                   (va - mf) / mf if mf else 0.0, lds, bc / lds if lds else 0.0, r["mfma_busy_frac"]))
with open(os.path.join(REPO, "profiles", "r03_dgcnn_instruction_mix.csv"), "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines))
