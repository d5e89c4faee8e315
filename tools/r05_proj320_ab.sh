# Round 5: PointNet++ sa2's 320-output projection: 256 columns on the bf16 pipe + 64 on the fp32 MFMA (default) against all 320
# on the fp32 MFMA's NT = 5 tiling (tuning key 5 = 59), one box, A B A B; then the dense-layer tests.
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  echo "pointnet2 fp32 NT=5 : $(timeout -k 10 200 python3 tools/bench_models.py --model pointnet2 --steps 8 --tune 5=59 2>&1 | tail -1 | cut -c1-230)"
  echo "pointnet2 256+64    : $(timeout -k 10 200 python3 tools/bench_models.py --model pointnet2 --steps 8 2>&1 | tail -1 | cut -c1-230)"
done
timeout -k 10 600 python3 -m pytest tests/test_hip_parity.py tests/test_pointnet2_gpu.py -x -q -m gpu 2>&1 | tail -8
