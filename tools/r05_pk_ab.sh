# Round 5, VERDICT item 1b: A/B of the two code generations (packed float32 allowed = round 4 | switched off = product), per
# family, alternating processes on ONE box (A B A B); bench_models prints coalitions/s and the per-slot kernel times.
#   bash tools/r05_pk_ab.sh > gpurun_out/r05ab/log.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ab; mkdir -p $O; cd $R
PK=$R/interpret_quality_amd/lib_packed_ab/libiq_hip.so
one() {  # $1 = tag, $2... = command
  tag=$1; shift
  for rep in 1 2; do
    echo "$tag packed   : $(IQ_LIBPATH=$PK "$@" 2>&1 | tail -1 | cut -c1-260)"
    echo "$tag no-packed: $("$@" 2>&1 | tail -1 | cut -c1-260)"
  done
}
hl() { python3 bench.py --steps 20 --repeats 3 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('headline %.0f coalitions/s, %.2f ms/step, chain launch %s ms' % (d['value'], d['ms_per_step'], r.get('avg_launch_ms')))"; }
for rep in 1 2; do
  echo "pointnet packed   : $(IQ_LIBPATH=$PK hl)"
  echo "pointnet no-packed: $(hl)"
done
one pointnet2 python3 tools/bench_models.py --model pointnet2 --mode shapley --steps 8
one pointconv python3 tools/bench_models.py --model pointconv --mode shapley --steps 8
one dgcnn python3 tools/bench_models.py --model dgcnn --mode interaction --steps 8
one gcnn python3 tools/bench_models.py --model gcnn --mode interaction --steps 8
