# usage (GPU box): bash tools/r04_pc_ab.sh -> gpurun_out/r04/pc_fused_ab.txt: PointConv step, sa1 as two kernels (5=31) vs fused
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for t in "5=31" "" "5=31" ""; do
  a=""; [ -n "$t" ] && a="--tune $t"
  echo "== pointconv tune=[$t]" >> $O/pc_fused_ab.txt
  python3 $R/tools/bench_models.py --model pointconv --steps 5 $a | tail -1 >> $O/pc_fused_ab.txt
done
cat $O/pc_fused_ab.txt
