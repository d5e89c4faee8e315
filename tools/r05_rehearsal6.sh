# Round 5: the multi-GPU entry points at SIX ranks on the one GPU of the box - the most the pool allows (its
# process guard kills a run with more than six GPU processes; the 8-way shard arithmetic, the padded all-gather and the pull
# queue at 8 ranks are rehearsed with gloo on the CPU: tests/test_launch_cpu.py).  IQ_REHEARSAL=1: every rank on cuda:0, gloo
# collectives.  No scaling number can come out of this (six processes share one card); what it shows is that the 6-way splits
# (217 poses -> 37,36,36,36,36,36; 300 pairs -> 50 each), the gathers and the pull queue run end to end and that the artefacts
# are the bits of a single process.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05reh; mkdir -p $O; cd $R
export IQ_REHEARSAL=1 PYTHONPATH=$R
echo "== bench.py --gpus 6 --scaling strong (one cloud's pose sweep + interaction setting sharded over 6 ranks)"
python3 bench.py --gpus 6 --scaling strong --steps 1 --warmup 0 --repeats 1 > $O/bench_strong_6rank.json 2> $O/bench_strong_6rank.err
tail -c 600 $O/bench_strong_6rank.json; echo
echo "== bench.py --gpus 6 (weak: every rank its own cloud, one all-gather of the logits per step)"
python3 bench.py --gpus 6 --steps 2 --warmup 1 --repeats 1 --perms 200 --strong-steps 0 --profile-steps 1 > $O/bench_weak_6rank.json 2> $O/bench_weak_6rank.err
tail -c 400 $O/bench_weak_6rank.json; echo
echo "== tools/sweep.py --gpus 6, all six models x modelnet10, reduced sizes, against ONE process: artefacts bitwise"
FLAGS="--datasets modelnet10 --synthetic --num_clouds 2 --num_samples_save 100 --num_pairs_random 5 --num_save_context_max 3"
rm -rf /tmp/reh6 /tmp/reh1; mkdir -p /tmp/reh6 /tmp/reh1
(cd /tmp/reh6 && python3 $R/tools/sweep.py --gpus 6 $FLAGS > $O/sweep_6rank.log 2> $O/sweep_6rank.err)
tail -1 $O/sweep_6rank.log > $O/sweep_6rank.json
grep "\[sweep\] phase\|\[sweep\] done" $O/sweep_6rank.log | cut -c1-220
(cd /tmp/reh1 && IQ_REHEARSAL=0 python3 $R/tools/sweep.py $FLAGS > $O/sweep_1rank.log 2> $O/sweep_1rank.err)
grep "\[sweep\] done" $O/sweep_1rank.log | cut -c1-220
python3 - <<'PY'
import glob, os
import numpy as np, torch
a, b = "/tmp/reh1/checkpoints", "/tmp/reh6/checkpoints"
n = bad = 0
for f in sorted(glob.glob(a + "/**/*", recursive=True)):
    if os.path.isdir(f) or f.endswith(".txt") or "/.sweep/" in f:
        continue
    g = f.replace(a, b); n += 1
    if not os.path.exists(g):
        print("missing", g); bad += 1; continue
    if f.endswith(".npy"):
        x, y = np.load(f), np.load(g); same = x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
    else:
        x, y = torch.load(f, map_location="cpu"), torch.load(g, map_location="cpu"); same = x.shape == y.shape and torch.equal(x, y)
    if not same:
        bad += 1; print("DIFF", f.replace(a, ""))
print("six ranks against one process: compared %d artefact files, %d different" % (n, bad))
raise SystemExit(1 if bad or n == 0 else 0)
PY
