# Round 5: PointConv's 16384 -> 1024 layer on B rows (split K, 512 k per split) on bf16x3 (default) against the fp32 MFMA's
# register-streaming kernel (tuning key 5 = 59), one box, A B A B; then PointConv's tests.
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  echo "pointconv fp32 split-K (5=59): $(timeout -k 10 200 python3 tools/bench_models.py --model pointconv --steps 8 --tune 5=59 2>&1 | tail -1 | cut -c1-230)"
  echo "pointconv bf16x3 split-K     : $(timeout -k 10 200 python3 tools/bench_models.py --model pointconv --steps 8 2>&1 | tail -1 | cut -c1-230)"
done
timeout -k 10 600 python3 -m pytest tests/test_pointconv_gpu.py tests/test_goldens_r2_gpu.py -x -q -m gpu 2>&1 | tail -4
