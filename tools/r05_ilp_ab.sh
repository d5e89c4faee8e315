# (A/B of a CLOSED experiment: apply tools/experiments/r05_grouped_staged_valu.patch first; see tools/experiments/README.md)
# Round 5: VALU phases written stage by stage over independent values (default) against value by value (round 4's order).
# Grouped bf16x3 kernels: tuning key 7 = 1 selects the old order inside ONE library; chain kernel: two builds (IQ_LIBPATH).
R=$GRAFT_REPO_ROOT; cd $R
OLD=$R/interpret_quality_amd/lib_chain_unstaged_ab/libiq_hip.so
for rep in 1 2 3; do
  for t in 1 0; do
    echo "pointnet2 7=$t: $(python3 tools/bench_models.py --model pointnet2 --mode shapley --steps 8 --tune 7=$t 2>&1 | tail -1 | cut -c1-200)"
  done
done
for rep in 1 2 3; do
  for t in 1 0; do
    echo "pointconv 7=$t: $(python3 tools/bench_models.py --model pointconv --mode shapley --steps 8 --tune 7=$t 2>&1 | tail -1 | cut -c1-200)"
  done
done
hl() { python3 bench.py --steps 20 --repeats 3 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0 --sustained-s 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('headline %.0f coalitions/s, %.2f ms/step, chain launch %.3f ms' % (d['value'], d['ms_per_step'], r.get('avg_launch_ms')))"; }
for rep in 1 2 3; do
  echo "pointnet value-by-value: $(IQ_LIBPATH=$OLD hl)"
  echo "pointnet staged        : $(hl)"
done
