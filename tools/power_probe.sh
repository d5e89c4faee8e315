# usage (GPU box): bash tools/power_probe.sh -> samples rocm-smi power / clocks while the headline step and the DGCNN probe run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
sample() {  # $1 = tag, $2 = pid to watch
  while kill -0 $2 2>/dev/null; do
    echo "$1 $(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' | cut -c1-900)" >> $O/samples.txt
    sleep 0.5
  done
}
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40 > $O/idle.txt
python3 $R/bench.py --steps 30 --repeats 8 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0 > $O/bench.json 2> $O/bench.err &
P=$!; sleep 14; sample headline $P; wait $P
IQ_BENCH_FP32_L3=1 python3 $R/bench.py --steps 30 --repeats 8 --cpu-baseline 0 --other-models 0 --eager-baseline 0 --traffic 0 --strong-steps 0 > $O/bench_fp32.json 2> $O/bench_fp32.err &
P=$!; sleep 14; sample headline_fp32_l3 $P; wait $P
python3 $R/tools/bench_models.py --model dgcnn --mode interaction --steps 40 > $O/dgcnn.log 2>&1 &
P=$!; sleep 10; sample dgcnn $P; wait $P
python3 - <<'PY'
import json, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04p")
rows = collections.defaultdict(list)
for ln in open(os.path.join(O, "samples.txt")):
    tag, t, js = ln.split(" ", 2)
    try:
        d = json.loads(js)
    except Exception:
        continue
    c = next(iter(d.values()))
    rows[tag].append(c)
for tag, cs in rows.items():
    keys = [k for k in cs[0] if "ower" in k or "sclk" in k.lower() or "mclk" in k.lower()]
    print(tag, len(cs), "samples")
    for k in keys:
        vals = [c.get(k) for c in cs]
        print("   ", k, vals[:3], "...", vals[len(vals) // 2])
PY
cut -c1-200 $O/bench.json | head -2; cut -c1-200 $O/bench_fp32.json | head -2; tail -1 $O/dgcnn.log | cut -c1-150; cat $O/idle.txt | head -30
