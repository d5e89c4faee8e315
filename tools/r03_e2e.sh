# usage (GPU box): bash tools/r03_e2e.sh [clouds]  -> gpurun_out/r03_e2e_times.txt
# Both pipelines (scripts/exp_shapley.sh + scripts/exp_interaction.sh) at the reference's sizes on synthetic clouds, per model:
# (a) the per-stage scripts one after the other, a process each; (b) tools/sweep.py, ONE process for everything (all six models).
set -e
n=${1:-1}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r03_e2e_times.txt; : > $out
for model in pointnet pointnet2 dgcnn gcnn_adv pointconv; do
  bash $R/tools/e2e_time.sh $model $n > /tmp/e2e_$model.txt 2>&1 || { tail -5 /tmp/e2e_$model.txt; exit 1; }
  echo "== $model: per-stage scripts, $n cloud(s)" >> $out; cat /tmp/e2e_$model.txt >> $out
  awk '{s+=$(NF-1)} END {printf "   total %d ms\n", s}' /tmp/e2e_$model.txt >> $out
done
W=/tmp/e2e_sweep; rm -rf $W; mkdir -p $W; cd $W
s=$(date +%s%N)
python $R/tools/sweep.py --synthetic --datasets modelnet10 --num_clouds $n > $W/sweep.log 2>&1 || { tail -20 $W/sweep.log; exit 1; }
e=$(date +%s%N)
echo "== tools/sweep.py: all six models, modelnet10, $n cloud(s), one process on one GPU" >> $out
grep "^\[sweep\]" $W/sweep.log >> $out
printf "   wall %d ms (including the one Python / PyTorch start-up)\n" $(( (e - s) / 1000000 )) >> $out
tail -12 $out
