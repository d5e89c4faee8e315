# Round 5: the bf16x3 kNN kernels of DGCNN - distance skeleton (tuning key 4 = 2: no selection; 4 = 1: queue appends without insert
# rounds; results invalid) for both arithmetics (5 = 22: fp32 MFMA).  The variants of profiles/r05_knn_bf3_ab.txt (k-steps ahead, two
# accumulators, queue slots, four-wave workgroups) were co-compiled behind 5 = 44 / 45 / 46 / 49 while they were measured with this
# script and are not in the tree.
R=$GRAFT_REPO_ROOT; cd $R
for t in "4=2,5=22" "4=2" "4=1,5=22" "4=1" "5=22" ""; do
  echo "tune [$t]: $(timeout -k 10 200 python3 tools/bench_models.py --model dgcnn --mode interaction --steps 8 --tune "$t" 2>&1 | tail -1 | cut -c1-200)"
done
