R=$GRAFT_REPO_ROOT; cd $R
for t in "" "5=45" "5=46" "5=49" "" "5=45" "5=46" "5=49" "4=2" "4=2,5=45" "4=2,5=46" "4=2,5=49"; do
  echo "tune [$t]: $(timeout -k 10 200 python3 tools/bench_models.py --model dgcnn --mode interaction --steps 8 --tune "$t" 2>&1 | tail -1 | cut -c1-200)"
done
