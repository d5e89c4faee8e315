# usage (GPU box): bash tools/e2e_time.sh <model> <num_clouds>   -> wall time of every stage of both pipelines on synthetic data
set -e
model=${1:-pointnet}; n=${2:-2}
R=$GRAFT_REPO_ROOT; W=/tmp/e2e_$model; rm -rf $W; mkdir -p $W; cd $W
for f in config.py final_*.py; do :; done
cp $R/config.py $R/final_*.py . ; ln -s $R/interpret_quality_amd interpret_quality_amd
common="--model=$model --dataset=modelnet10 --synthetic --num_clouds $n"
t() { local s=$(date +%s%N); "$@" > $W/last.log 2>&1 || { tail -5 $W/last.log; exit 1; }; local e=$(date +%s%N); printf "%-52s %8d ms\n" "$2" $(( (e - s) / 1000000 )); }
t python final_shapley_value.py $common
t python final_trans_center_enum_all.py $common
t python final_rotate_center_enum_all.py $common
t python final_scale_center_enum_all.py $common
t python final_smoothness_center_enum_all.py $common
t python final_gen_pair.py $common
t python final_point_binary_interaction_logits.py $common
t python final_cal_interactions.py $common
du -sh checkpoints | cut -f1
