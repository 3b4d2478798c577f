#!/bin/bash
# In-session A/B of the element route's variants (box-to-box spread is ~10 %: only numbers of one session compare).
#   tools/r03_matrix.sh OUTDIR "T N" VARIANT...      a VARIANT is a comma-separated list of VAR=value
out=$1; tn=$2; shift 2
mkdir -p $out; L=$out/matrix.log
for rep in 1 2; do
  for v in "$@"; do
    echo "== $tn $v" >> $L
    env $(echo $v | tr ',' ' ') python tools/elem_bench.py $tn 30 2>&1 | grep -v amdgpu.ids | sed -e 's/n_int=.*apex [0-9/]*//' >> $L
  done
done
