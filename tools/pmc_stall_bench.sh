# usage (GPU box): bash tools/pmc_stall_bench.sh <tag> -> gpurun_out/pmc_stall_<tag>/{a,b}: two counter passes of bench.py (headline step only)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stall_$tag/a -- $B > $R/gpurun_out/pmc_stall_$tag.a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stall_$tag/b -- $B > $R/gpurun_out/pmc_stall_$tag.b.log 2>&1
echo done
