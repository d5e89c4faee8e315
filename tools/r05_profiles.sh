# usage (GPU box): bash tools/r05_profiles.sh  -> gpurun_out/r05f/*: the rocprofv3 runs and bench lines the round-5 numbers in DESIGN.md come from
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --min-region-s 0 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $B > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/bench_pmc -- $B > $O/bench_pmc.log 2>&1
for m in pointnet2 dgcnn gcnn pointconv; do
  mode=shapley; [ $m = dgcnn ] && mode=interaction; [ $m = gcnn ] && mode=interaction
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 3 > $O/stats_$m.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 1 > $O/pmc_$m.log 2>&1
  grep coalitions $O/stats_$m.log
done
# the headline line itself (the driver's command) and the configs[4] sweep line on one GPU (the 6-rank rehearsals: tools/r05_rehearsal6.sh)
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --scaling sweep --sweep-datasets modelnet10 --sweep-clouds 1 > $O/bench_sweep_1gpu.json 2> $O/bench_sweep_1gpu.err
tail -c 600 $O/bench.json; echo; tail -c 400 $O/bench_sweep_1gpu.json
