# Round 5 against the end of round 4 (commit b4d5fcb) on ONE box: the round-4 tree is extracted and built under build/r4_tree
# (git archive b4d5fcb --prefix=r4_tree/ | tar -x -C build; python -m interpret_quality_amd.build there), each tree runs its own
# tools/bench_models.py / bench.py, alternating.  Board-to-board spread (the kernels are power-bound) makes cross-round numbers from
# different boxes fuzzy; this table is not.
R=$GRAFT_REPO_ROOT; O=$R/build/r4_tree
one() {  # dir model mode
  (cd $1 && timeout -k 10 200 python3 tools/bench_models.py --model $2 --mode $3 --steps 8 2>&1 | tail -1 | cut -c1-70)
}
for m in pointnet2:shapley pointconv:shapley dgcnn:interaction gcnn:interaction; do
  mod=${m%%:*}; mode=${m##*:}
  for rep in 1 2; do
    echo "r4 $(one $O $mod $mode)"
    echo "r5 $(one $R $mod $mode)"
  done
done
B="bench.py --repeats 1 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0"
for rep in 1 2; do
  echo "r4 headline $(cd $O && timeout -k 10 300 python3 $B 2>/dev/null | python3 -c 'import sys,json; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(b["value"]), round(b["ms_per_step"],2))')"
  echo "r5 headline $(cd $R && timeout -k 10 300 python3 $B --sustained-s 0 2>/dev/null | python3 -c 'import sys,json; b=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(round(b["value"]), round(b["ms_per_step"],2))')"
done
