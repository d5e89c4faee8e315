# usage (GPU box): bash tools/pmc_stall.sh <tag> <bench_models args...>  -> gpurun_out/pmc_stall_<tag>/{a,b} (two rocprofv3 counter passes,
# kernel trace only): where the waves of a model's kernels spend their cycles (issue / wait / MFMA busy / LDS / VMEM)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stall_$tag/a -- python3 $R/tools/bench_models.py "$@" --steps 1 > $R/gpurun_out/pmc_stall_$tag.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stall_$tag/b -- python3 $R/tools/bench_models.py "$@" --steps 1 > $R/gpurun_out/pmc_stall_$tag.b.log 2>&1
grep coalitions $R/gpurun_out/pmc_stall_$tag.a.log
