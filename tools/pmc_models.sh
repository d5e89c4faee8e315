# usage (GPU box): bash tools/pmc_models.sh <model> <mode>  -> gpurun_out/pmc_<model>/ (rocprofv3 counter pass, kernel trace only)
set -e
m=$1; mode=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 1 > $R/gpurun_out/pmc_$m.log 2>&1
grep coalitions $R/gpurun_out/pmc_$m.log
