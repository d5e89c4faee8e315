# Round 5, second GPU step: (1) ignored-half patterns of packed float32 operands, alone; (2) finer bisect of the smoothness clone
# beside the bf16x3 chain kernel; (3) issue priority by phase (tuning key 7) in the grouped bf16x3 kernels and the chain kernel.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
echo "== 1. ignored half of a packed operand (alone)"
tools/micro/pk_victim.bin 3 1 64 20000
echo "== 2. variances() parts beside the bf16x3 chain kernel"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 60 > $O/neighbour5.log 2>&1 &
NB=$!
sleep 20
for v in 100 101 104 105; do tools/micro/smooth_victim.bin $v 6; done
wait $NB; echo "neighbour exit $?"
echo "== 3. issue priority by phase (7 = 0 off | 1 MFMA loops high | 2 VALU phases high)"
for rep in 1 2; do
  for t in 0 1 2; do
    echo "pointnet2 7=$t: $(python3 tools/bench_models.py --model pointnet2 --mode shapley --steps 8 --tune 7=$t 2>&1 | tail -1 | cut -c1-200)"
  done
done
for rep in 1 2; do
  for t in 0 1 2; do
    echo "pointconv 7=$t: $(python3 tools/bench_models.py --model pointconv --mode shapley --steps 8 --tune 7=$t 2>&1 | tail -1 | cut -c1-200)"
  done
done
python3 tools/ab_chain.py --key 7 --values 0,1 --rounds 7 2>&1 | tail -8
