#!/bin/bash
# What would ONE round trip instead of two in the staging step of the patch kernel save?  Ablation build, FEP_STAGE_NOCHAIN=1
# (gather addresses independent of the loaded ids; results wrong, timing only), phase clocks beside.
out=gpurun_out/r4nochain; mkdir -p $out; rm -f $out/*.log
export TMPDIR=/tmp
C=$PWD/fem-elastoplasticity_amd/csrc
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_PHASE_CLK=1 "$t"
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_PHASE_CLK=1 FEP_STAGE_NOCHAIN=1 "$t"
done
done
grep -o "^== .*\|step [0-9.]* ms\|'element': [0-9.]*\|phase3): [0-9 ]* sum [0-9]*" $out/elem_bench.log | paste - - - - | sed 's/FEP_LIB_PATH=[^ ]*libfep_hip_//'
