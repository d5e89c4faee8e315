# Round 5: where conv5's pooled bf16x3 GEMM spends its time - timing probes (results wrong): 5 = 94 no split arithmetic when the
# activations are staged, 5 = 95 no pooling epilogue.  GCNN: conv5 is 3/4 of its step.  (The eight-wave workgroup variant measured with
# this script - 29.5 ms against 29.1 - is gone again: iq_linear.hip, note in pn_gemm_bf3_kernel's epilogue.)
R=$GRAFT_REPO_ROOT; cd $R
for t in "" "5=94" "5=95" "" "5=94" "5=95"; do
  echo "gcnn tune [$t]: $(timeout -k 10 200 python3 tools/bench_models.py --model gcnn --mode interaction --steps 8 --tune "$t" 2>&1 | tail -1 | cut -c1-200)"
done
