#!/bin/bash
# P1 benchmark over plan / kernel variants: `tools/bench_variants.sh OUTDIR "SEGS..." "FUSED..."` (one bench.py run each)
out=${1:-gpurun_out/variants}; mkdir -p $out
for segs in ${2:-1 2 4}; do for fused in ${3:-all off}; do
  FEP_P1_SEGS=$segs FEP_P1_FUSED=$fused python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/b_s${segs}_${fused}.json 2> $out/b_s${segs}_${fused}.err || echo "FAILED $segs $fused"
  python - <<PY
import json
j=json.load(open("$out/b_s${segs}_${fused}.json"))
print("segs=$segs fused=$fused  full: %.1f us/step (kernels %s)  kf: %.1f us/step (%s)  frac %.3f" % (j['ms_per_step']*1e3, {k:round(v*1e3,1) for k,v in j['roofline']['kernels_ms'].items()}, j['kf_only']['ms_per_step']*1e3, {k:round(v*1e3,1) for k,v in j['kf_only']['kernels_ms'].items()}, j['roofline']['frac']))
PY
done; done
