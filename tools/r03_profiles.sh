# usage (GPU box): bash tools/r03_profiles.sh  -> gpurun_out/r03/*: the rocprofv3 runs the round-3 numbers in DESIGN.md come from
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $B > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/bench_pmc -- $B > $O/bench_pmc.log 2>&1
for m in pointnet2 dgcnn gcnn pointconv; do
  mode=shapley; [ $m = dgcnn ] && mode=interaction; [ $m = gcnn ] && mode=interaction
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 3 > $O/stats_$m.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 1 > $O/pmc_$m.log 2>&1
  grep coalitions $O/stats_$m.log
done
# PointNet++ sa1 gather: region-reduced tables against the member walk (tuning key 5 = 21), fabric-side fetch counters
for v in reg walk; do
  t=""; [ $v = walk ] && t="--tune 5=21"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/gather_fetch_$v -- python3 $R/tools/bench_models.py --model pointnet2 --steps 2 $t > $O/gather_fetch_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/gather_write_$v -- python3 $R/tools/bench_models.py --model pointnet2 --steps 2 $t > $O/gather_write_$v.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/gather_tcc_$v -- python3 $R/tools/bench_models.py --model pointnet2 --steps 2 $t > $O/gather_tcc_$v.log 2>&1
done
python3 $R/tools/bench_models.py --model dgcnn --mode interaction --tune 4=3 --steps 1 > $O/knn_counters.log 2>&1
python3 $R/tools/bench_models.py --model dgcnn --mode interaction --tune 5=20 > $O/dgcnn_no_refine.log 2>&1
python3 $R/tools/bench_models.py --model dgcnn --mode interaction > $O/dgcnn_refine.log 2>&1
for f in knn_counters dgcnn_no_refine dgcnn_refine; do tail -n 1 $O/$f.log; done
