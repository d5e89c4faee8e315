# Round 5, third GPU step: packed against scalar loop 1 inside ONE launch (tools/micro/smooth_victim.hip, variants 200 / 201)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
echo "== alone"
tools/micro/smooth_victim.bin 200 4
echo "== beside the bf16x3 chain kernel"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 45 > $O/neighbour6.log 2>&1 &
NB=$!
sleep 20
tools/micro/smooth_victim.bin 200 7
tools/micro/smooth_victim.bin 201 7
wait $NB; echo "neighbour exit $?"
