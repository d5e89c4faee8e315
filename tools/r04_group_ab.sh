# A/B of the bf16x3 grouped kernels: blocks per workgroup (tuning key 6) for PointNet++; kernel trace of both families
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in 0 16 32 48 64; do
  echo "pointnet2 tune 6=$t: $(python3 $R/tools/bench_models.py --model pointnet2 --mode shapley --steps 5 --tune 6=$t 2>&1 | tail -1 | cut -c1-200)"
done
for m in pointnet2 pointconv; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$m -- python3 $R/tools/bench_models.py --model $m --mode shapley --steps 3 > $O/stats_$m.log 2>&1
  f=$(find $O/stats_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m"; head -9 $f | cut -d, -f1-5 | sed 's/(anonymous namespace):://g' | cut -c1-150
done
