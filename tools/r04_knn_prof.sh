# usage (GPU box): bash tools/r04_knn_prof.sh <tag> [bench_models args] -> gpurun_out/r04/knn_<tag>_*: DGCNN interaction step, kernel trace + two SQ counter passes
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BM="python3 $R/tools/bench_models.py --model dgcnn --mode interaction"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/knn_${tag}_stats -- $BM --steps 3 "$@" > $O/knn_${tag}_stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/knn_${tag}_a -- $BM --steps 1 "$@" > $O/knn_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/knn_${tag}_b -- $BM --steps 1 "$@" > $O/knn_${tag}_b.log 2>&1
python3 $R/tools/pmc_summarise.py $O/knn_${tag}_a $O/knn_${tag}_b > $O/knn_${tag}_summary.csv
head -8 $O/knn_${tag}_stats/*/*kernel_stats.csv | cut -c1-140
tail -1 $O/knn_${tag}_stats.log
