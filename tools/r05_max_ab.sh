# Round 5: pooling maxima without the canonicalising v_max (product) against the previous build (lib_prev_ab), one box, A B A B
R=$GRAFT_REPO_ROOT; cd $R
OLD=$R/interpret_quality_amd/lib_prev_ab/libiq_hip.so
hl() { python3 bench.py --steps 20 --repeats 3 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0 --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('headline %.0f coalitions/s, %.2f ms/step, chain launch %.3f ms' % (d['value'], d['ms_per_step'], r.get('avg_launch_ms')))"; }
for rep in 1 2 3; do
  echo "pointnet previous : $(IQ_LIBPATH=$OLD hl)"
  echo "pointnet transposed: $(hl)"
done
for rep in 1 2 3; do
  echo "pointnet2 previous: $(IQ_LIBPATH=$OLD python3 tools/bench_models.py --model pointnet2 --mode shapley --steps 8 2>&1 | tail -1 | cut -c1-200)"
  echo "pointnet2 same     : $(python3 tools/bench_models.py --model pointnet2 --mode shapley --steps 8 2>&1 | tail -1 | cut -c1-200)"
done
python3 tools/chain_repro.py --seconds 5 | tail -1
