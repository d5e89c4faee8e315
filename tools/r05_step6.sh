# Round 5: packed operand forms (op_sel cross reads, ignored halves) on REGISTERS beside the bf16x3 chain kernel: still clean?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 40 > $O/neighbour8.log 2>&1 &
NB=$!
sleep 20
tools/micro/pk_victim.bin 3 10 64 20000 1 | tail -8
wait $NB; echo "neighbour exit $?"
