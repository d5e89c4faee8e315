# Round 5: the ten single-modifier operand forms of packed float32 on REGISTERS (tools/micro/pk_victim.hip variant 3), alone and
# beside the bf16x3 chain kernel of a second process
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
echo "== alone"
tools/micro/pk_victim.bin 3 6 64 20000
echo "== beside the bf16x3 chain kernel (second process)"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 60 > $O/neighbour9.log 2>&1 &
NB=$!
sleep 20
tools/micro/pk_victim.bin 3 12 64 20000
tools/micro/pk_victim.bin 3 8 64 200
wait $NB; echo "neighbour exit $?"
echo "== beside the chain kernel with layer 3 on the fp32 MFMA"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet_fp32_l3 --seconds 40 > $O/neighbour10.log 2>&1 &
NB=$!
sleep 20
tools/micro/pk_victim.bin 3 8 64 20000
wait $NB; echo "neighbour exit $?"
