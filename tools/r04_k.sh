#!/bin/bash
# k2: plastic strain fetched beside the staging chain (unconditional loads), 15-node gradients in one round trip — A/B
out=gpurun_out/r4k; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
C=$PWD/fem-elastoplasticity_amd/csrc
FEP_LIB_PATH=$C/libfep_hip_k2.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/pytest_parity_k2.log 2>&1; echo "pytest k2 rc=$?"; tail -3 $out/pytest_parity_k2.log
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run FEP_LIB_PATH=$C/libfep_hip_r03.so "$t"
run X=cur "$t"
run FEP_LIB_PATH=$C/libfep_hip_k2.so "$t"
done
done
for i in 1 2; do
run X=cur "P2 1414 10 random"
run FEP_LIB_PATH=$C/libfep_hip_k2.so "P2 1414 10 random"
run X=cur "P2 708 30 bands kf"
run FEP_LIB_PATH=$C/libfep_hip_k2.so "P2 708 30 bands kf"
done
cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  "
