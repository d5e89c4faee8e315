# Round 5: conv5's pooling epilogue with the column maximum taken over the raw sums (one add + one max per value) against the build
# before (closed experiment: needs that build as interpret_quality_amd/lib_prev_ab/), one box, A B A B; GCNN (conv5 = 3/4 of its step), DGCNN.
R=$GRAFT_REPO_ROOT; cd $R
OLD=$R/interpret_quality_amd/lib_prev_ab/libiq_hip.so
for m in gcnn dgcnn; do
  for rep in 1 2 3; do
    echo "$m previous: $(IQ_LIBPATH=$OLD timeout -k 10 200 python3 tools/bench_models.py --model $m --mode interaction --steps 8 2>&1 | tail -1 | cut -c1-210)"
    echo "$m new     : $(timeout -k 10 200 python3 tools/bench_models.py --model $m --mode interaction --steps 8 2>&1 | tail -1 | cut -c1-210)"
  done
done
timeout -k 10 600 python3 -m pytest tests/test_dgcnn_gpu.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4
