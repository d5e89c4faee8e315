# usage (GPU box): bash tools/r04_gemm_ab.sh -> gpurun_out/r04/gemm_ab.txt: PointNet++ and PointConv steps with round 3's GEMM tilings (5=30) and round 4's
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
for m in pointnet2 pointconv; do
  for t in "5=30" ""; do
    a=""; [ -n "$t" ] && a="--tune $t"
    echo "== $m tune=[$t]" >> $O/gemm_ab.txt
    python3 $R/tools/bench_models.py --model $m --steps 5 $a | tail -1 >> $O/gemm_ab.txt
  done
done
cat $O/gemm_ab.txt
