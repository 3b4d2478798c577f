#!/bin/bash
# matrix of the last kernel variants: s = skewed phase-3 image, h = plastic strain beside the staging chain, u = staging loads
# unconditional; all with the 15-node gradients in one round trip; cur = the committed kernel
out=gpurun_out/r4m; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
C=$PWD/fem-elastoplasticity_amd/csrc
FEP_LIB_PATH=$C/libfep_hip_k_hus.so timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/pytest_parity_hus.log 2>&1; echo "pytest hus rc=$?"; tail -3 $out/pytest_parity_hus.log
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run X=cur "$t"
for v in s h u hs hus; do run FEP_LIB_PATH=$C/libfep_hip_k_$v.so "$t"; done
done
done
cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  "
