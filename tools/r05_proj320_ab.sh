# Round 5: dense layers newly on bf16x3 - PointNet++ sa2's 320-output projection (256 columns there + 64 on the fp32 MFMA) and the
# sa3 input layers whose inputs are no multiple of 32 (PointNet++ 643 -> 648 columns, PointConv 259 -> 264) - against the fp32 MFMA
# for exactly these layers (tuning key 5 = 59), one box, A B A B; then the dense-layer and model tests.
R=$GRAFT_REPO_ROOT; cd $R
for m in pointnet2 pointconv; do
  for rep in 1 2 3; do
    echo "$m fp32 (5=59): $(timeout -k 10 200 python3 tools/bench_models.py --model $m --steps 8 --tune 5=59 2>&1 | tail -1 | cut -c1-230)"
    echo "$m bf16x3     : $(timeout -k 10 200 python3 tools/bench_models.py --model $m --steps 8 2>&1 | tail -1 | cut -c1-230)"
  done
done
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_pointnet2_gpu.py tests/test_pointconv_gpu.py -x -q -m gpu 2>&1 | tail -8
