set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in pointnet2 dgcnn gcnn pointconv; do
  mode=shapley; [ $m = dgcnn ] && mode=interaction
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$m -- python3 $R/tools/bench_models.py --model $m --mode $mode --steps 3 > $R/gpurun_out/prof_$m.log 2>&1
  echo "$m done"
done
