# Round 5: bisect the shared-GPU effect with the reduced clone of the smoothness loop (tools/micro/smooth_victim.hip).  GPU box.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
V=$R/tools/micro/smooth_victim.bin
echo "== alone"
$V 7 5; $V 0 5
echo "== beside the bf16x3 chain kernel (second process, round-4 code generation)"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 95 > $O/neighbour3.log 2>&1 &
NB=$!
sleep 20
for v in 7 0 1 2 4 7; do $V $v 8; done
$V 7 8 32 2000
wait $NB; echo "neighbour exit $?"
echo "== beside the chain kernel with layer 3 on the fp32 MFMA (round 4: no effect)"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet_fp32_l3 --seconds 40 > $O/neighbour4.log 2>&1 &
NB=$!
sleep 20
$V 7 8
wait $NB; echo "neighbour exit $?"
