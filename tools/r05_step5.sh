# Round 5: last bisect of the in-launch victim: idle cycles after the LDS data arrived (202), points through global memory (203)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 50 > $O/neighbour7.log 2>&1 &
NB=$!
sleep 20
tools/micro/smooth_victim.bin 200 5 | tail -1
tools/micro/smooth_victim.bin 202 5 | tail -3
tools/micro/smooth_victim.bin 203 5 | tail -3
tools/micro/smooth_victim.bin 200 5 64 400 | tail -1
wait $NB; echo "neighbour exit $?"
echo "== chain kernel layer-3 timing probes (results wrong): 0 = product, 91 no weight loads, 92 A terms read once per pass, 93 both"
python3 tools/ab_chain.py --key 5 --values 0,91,92,93 --rounds 5 2>&1 | tail -5
