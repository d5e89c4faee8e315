# usage (GPU box): bash tools/r04_profiles_a.sh  -> gpurun_out/r04/*: round 4's first measurements, BEFORE any kernel change
#   (1) PointConv sa1 pair (pc_tab_group_kernel + the 2048 -> 128 layer behind it): fabric-side bytes and L2 hit rate
#   (2) PointNet++ / PointConv / DGCNN: where the waves of pn_gemm_lds_kernel spend their cycles (the same template at 0.72 / 0.61 / 0.85)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BM="python3 $R/tools/bench_models.py"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  t=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pc_$t -- $BM --model pointconv --steps 2 > $O/pc_$t.log 2>&1
done
for m in pointnet2 pointconv; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/stall_a_$m -- $BM --model $m --steps 1 > $O/stall_a_$m.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/stall_b_$m -- $BM --model $m --steps 1 > $O/stall_b_$m.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -- $BM --model $m --steps 2 > $O/trace_$m.log 2>&1
done
python3 $R/tools/pmc_per_dispatch.py $O/pc_FETCH_SIZE $O/pc_WRITE_SIZE $O/pc_TCC_HIT_sum > $O/pc_traffic_summary.csv
python3 $R/tools/pmc_summarise.py $O/stall_a_pointnet2 $O/stall_b_pointnet2 > $O/stall_pointnet2_summary.csv
python3 $R/tools/pmc_summarise.py $O/stall_a_pointconv $O/stall_b_pointconv > $O/stall_pointconv_summary.csv
grep coalitions $O/*.log | tail -8
