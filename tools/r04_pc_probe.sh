# usage (GPU box): bash tools/r04_pc_probe.sh -> gpurun_out/r04/probe_*.csv: the fused sa1 kernel's timing probes (tuning key 5 = 32..36) under the kernel trace and one counter pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for k in 34 33; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/probe_$k -- python3 $R/tools/bench_models.py --model pointconv --steps 1 --tune 5=$k > $O/probe_$k.log 2>&1
  python3 $R/tools/pmc_summarise.py $O/probe_$k | grep "kernel\|fused" | cut -d, -f1-3,8,12,14,15,16,17 > $O/probe_$k.csv
  echo "5=$k"; cat $O/probe_$k.csv
done
