# usage (GPU box): bash tools/r03_profiles_dgcnn.sh  -> gpurun_out/r03/{stats,pmc}_dgcnn: the DGCNN part of tools/r03_profiles.sh again
# (after the kNN selection changed)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
m=dgcnn; mode=interaction
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 3 > $O/stats_$m.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 1 > $O/pmc_$m.log 2>&1
grep coalitions $O/stats_$m.log
python3 $R/tools/bench_models.py --model dgcnn --mode interaction --tune 4=3 --steps 1 > $O/knn_counters.log 2>&1
python3 $R/tools/bench_models.py --model dgcnn --mode interaction --tune 5=20 > $O/dgcnn_no_refine.log 2>&1
python3 $R/tools/bench_models.py --model dgcnn --mode interaction > $O/dgcnn_refine.log 2>&1
for f in knn_counters dgcnn_no_refine dgcnn_refine; do tail -n 1 $O/$f.log; done
