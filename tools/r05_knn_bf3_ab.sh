# Round 5: DGCNN's feature-space kNN distances on the bf16 matrix pipe (bf16x3) against the fp32 MFMA kernels (tuning key 5 = 22),
# one box, A B A B; then DGCNN's parity tests.
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  echo "dgcnn fp32 kNN : $(timeout -k 10 200 python3 tools/bench_models.py --model dgcnn --mode interaction --steps 8 --tune 5=22 2>&1 | tail -1 | cut -c1-260)"
  echo "dgcnn bf16x3   : $(timeout -k 10 200 python3 tools/bench_models.py --model dgcnn --mode interaction --steps 8 2>&1 | tail -1 | cut -c1-260)"
done
timeout -k 10 600 python3 -m pytest tests/test_dgcnn_gpu.py -x -q -m gpu 2>&1 | tail -15
