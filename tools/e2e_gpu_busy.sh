# usage (GPU box): bash tools/e2e_gpu_busy.sh <model>  -> wall time of the interaction-logits stage of one cloud and the sum of its kernel times
set -e
model=${1:-pointnet2}
R=$GRAFT_REPO_ROOT; W=/tmp/e2e_busy_$model; rm -rf $W; mkdir -p $W; cd $W
cp $R/config.py $R/final_*.py . ; ln -s $R/interpret_quality_amd interpret_quality_amd
common="--model=$model --dataset=modelnet10 --synthetic --num_clouds 1"
python final_shapley_value.py $common > /dev/null 2>&1
python final_rotate_center_enum_all.py $common > /dev/null 2>&1
python final_gen_pair.py $common > /dev/null 2>&1
export TMPDIR=/tmp
s=$(date +%s%N)
rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof -- python3 final_point_binary_interaction_logits.py $common > $W/logits.log 2>&1
e=$(date +%s%N)
python3 - <<PY
import csv, glob
f = glob.glob("$W/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e9
print("$model interaction-logits stage: wall %.1f s (under rocprofv3, incl. start-up), kernels %.1f s" % (($e - $s) / 1e9, tot))
for r in rows[:6]:
    print("   %-70s %6.2f s  %s %%" % (r["Name"][:70], float(r["TotalDurationNs"]) / 1e9, r["Percentage"]))
PY
