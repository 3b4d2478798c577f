#!/bin/bash
# Upper bound of an overlapped fix-up (ablation build, FEP_FIX_SIDE=1: fixup_kernel on a side stream, not ordered behind the element kernel)
out=gpurun_out/r4side; mkdir -p $out; rm -f $out/elem_bench.log
export TMPDIR=/tmp
C=$PWD/fem-elastoplasticity_amd/csrc
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "P4 354 30" "Q1 708 30"; do
run FEP_LIB_PATH=$C/libfep_hip_abl.so "$t"
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_FIX_SIDE=1 "$t"
done
done
cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  File"
