# Round 5: -fno-honor-nans also for iq_linear.hip (conv5 + pool, dense layers) and iq_dgcnn.hip (EdgeConv gather maxima), against the
# build before (lib_prev_ab2: only iq_pointnet.hip), one box, A B A B
R=$GRAFT_REPO_ROOT; cd $R
OLD=$R/interpret_quality_amd/lib_prev_ab2/libiq_hip.so
for m in dgcnn gcnn; do
  for rep in 1 2 3; do
    echo "$m previous: $(IQ_LIBPATH=$OLD python3 tools/bench_models.py --model $m --mode interaction --steps 8 2>&1 | tail -1 | cut -c1-210)"
    echo "$m no-nans : $(python3 tools/bench_models.py --model $m --mode interaction --steps 8 2>&1 | tail -1 | cut -c1-210)"
  done
done
for rep in 1 2; do
  echo "pointconv previous: $(IQ_LIBPATH=$OLD python3 tools/bench_models.py --model pointconv --mode shapley --steps 8 2>&1 | tail -1 | cut -c1-120)"
  echo "pointconv no-nans : $(python3 tools/bench_models.py --model pointconv --mode shapley --steps 8 2>&1 | tail -1 | cut -c1-120)"
done
