# usage (GPU box): bash tools/two_rank_check.sh <model>  -> every stage of both pipelines with 1 process and with 2 ranks
# (IQ_REHEARSAL=1: both on cuda:0, gloo), then a bitwise comparison of all artefacts.
set -e
m=$1
R=$GRAFT_REPO_ROOT; W=/tmp/tr_$m; rm -rf $W; mkdir -p $W/one $W/two
export PYTHONPATH=$R IQ_REHEARSAL=1
common="--model=$m --dataset=modelnet10 --synthetic --num_clouds 1"
k=0
for s in "final_shapley_value.py --num_samples_save 100" "final_rotate_center_enum_all.py" "final_smoothness_center_enum_all.py" "final_gen_pair.py --num_pairs_random 4 --num_save_context_max 3" "final_point_binary_interaction_logits.py" "final_cal_interactions.py"; do
  (cd $W/one && python $R/$s $common > log.txt 2>&1) || { echo "ONE FAILED: $s"; tail -5 $W/one/log.txt; exit 1; }
  k=$((k+1))
  (cd $W/two && python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700+k)) $R/$s $common > log.txt 2>&1) || { echo "TWO FAILED: $s"; grep -v "^W\|warn" $W/two/log.txt | tail -15; exit 1; }
done
python - <<PY
import numpy as np, torch, glob, os
a="$W/one/checkpoints"; b="$W/two/checkpoints"
bad=0; n=0
for f in sorted(glob.glob(a+"/**/*", recursive=True)):
    if os.path.isdir(f) or f.endswith("log.txt") or f.endswith(".txt"): continue
    g=f.replace(a,b); n+=1
    if not os.path.exists(g): print("missing", g); bad+=1; continue
    if f.endswith(".npy"):
        x,y=np.load(f),np.load(g); same = x.shape==y.shape and np.array_equal(x,y,equal_nan=True)
    else:
        x,y=torch.load(f,map_location="cpu"),torch.load(g,map_location="cpu"); same = x.shape==y.shape and torch.equal(x,y)
    if not same:
        bad+=1
        d = np.abs(np.asarray(x,dtype=np.float64)-np.asarray(y,dtype=np.float64)).max() if x.shape==y.shape else -1
        print("DIFF", f.replace(a,""), d)
print("$m: compared", n, "files,", bad, "different")
PY
