# Round 5, VERDICT item 1c: which ingredient makes a neighbour process change a kernel's bits?  (GPU box)
#   bash tools/r05_pk_victim.sh > gpurun_out/r05pk/log.txt
# 1. the minimal victim (registers only, packed float32 against its scalar twin, tools/micro/pk_victim.hip) alone and beside
#    the PointNet chain kernel of ROUND 4's code generation (lib_packed_ab) in a second process;
# 2. positive control: round 4's victim (the smoothness enumeration built WITH packed float32) beside the same neighbour;
# 3. the product build of the smoothness kernel (no packed float32) beside the same neighbour;
# 4. the chain kernel against itself (same launch repeated), both code generations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05pk; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
V=$R/tools/micro/pk_victim.bin
echo "== 1a. minimal victim alone"
for v in 0 1 2; do $V $v 6 64; done
$V 0 6 1024; $V 1 6 1024
echo "== 1b. minimal victim beside the bf16x3 chain kernel (second process, round-4 code generation)"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 75 > $O/neighbour1.log 2>&1 &
NB=$!
sleep 20
for v in 0 1 2; do $V $v 8 64; done
$V 0 8 1024; $V 1 8 1024; $V 0 8 32 400000
wait $NB; echo "neighbour exit $?"
echo "== 1c. minimal victim beside the bf16x3 chain kernel (product build: no packed float32 in the neighbour either)"
python3 tools/shared_gpu_determinism.py --role load --load pointnet --seconds 45 > $O/neighbour2.log 2>&1 &
NB=$!
sleep 20
$V 0 8 64; $V 1 8 64; $V 1 8 1024
wait $NB; echo "neighbour exit $?"
echo "== 2. positive control: smoothness kernel WITH packed float32 (round-4 code generation, both processes)"
IQ_LIBPATH=$PK python3 tools/shared_gpu_determinism.py --load pointnet --seconds 20 2>&1 | tail -6
echo "== 3. product build (no packed float32 anywhere), same experiment"
python3 tools/shared_gpu_determinism.py --load pointnet --seconds 20 2>&1 | tail -6
echo "== 4. the chain kernel against itself"
IQ_LIBPATH=$PK python3 tools/chain_repro.py --seconds 12 2>&1 | tail -1
python3 tools/chain_repro.py --seconds 12 2>&1 | tail -1
