#!/bin/bash
# Side classification (blocks that share a 64-byte chunk with a foreign piece go through the side buffer; the fix-up writes whole
# chunks): parity on the new library, then in-session A/B against the library of the commit before (csrc/libfep_hip_prev.so) and
# the closure rounds 0..3 (ablation build, FEP_SIDE_ROUNDS); K and F compared by SHA-1
out=gpurun_out/r4sidefix; mkdir -p $out; rm -f $out/*.log
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
FEP_VALIDATE_PLAN=1 timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/parity.log 2>&1; rc=$?; tail -3 $out/parity.log
[ $rc -ne 0 ] && exit $rc
C=$PWD/fem-elastoplasticity_amd/csrc
export FEP_BENCH_HASH=1
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run FEP_LIB_PATH=$C/libfep_hip_prev.so "$t"
run X=product "$t"
for k in 0 2 3; do run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_SIDE_ROUNDS=$k "$t"; done
done
done
run FEP_LIB_PATH=$C/libfep_hip_prev.so "P2 1414 10 random"
run X=product "P2 1414 10 random"
grep -o "^== [A-Za-z_=0-9/.]* [A-Z_=0-9]*\|^[PQ][124] N=[0-9]*\|step [0-9.]* ms\|'element': [0-9.]*, 'csr': [0-9.]*\|sha1.*" $out/elem_bench.log | paste - - - - - | sed 's/FEP_LIB_PATH=[^ ]*libfep_hip_//'
