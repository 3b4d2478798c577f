#!/bin/bash
# one-patch-per-workgroup kernel with every phase-3 descriptor fetched in front of phase 2: parity, A/B against round 3
out=gpurun_out/r4i; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/pytest_parity.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_parity.log
C=$PWD/fem-elastoplasticity_amd/csrc
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run FEP_LIB_PATH=$C/libfep_hip_r03.so "$t"
run X=new "$t"
done
done
run FEP_LIB_PATH=$C/libfep_hip_r03.so "P2 1414 10 random"
run X=new "P2 1414 10 random"
run FEP_LIB_PATH=$C/libfep_hip_r03.so "P2 1414 10 random"
run X=new "P2 1414 10 random"
for t in "P2 708 5" "Q2 708 5" "Q1 708 5" "P4 354 5"; do
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_PHASE_CLK=1 "$t"
done

cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  "

bash tools/r04_h.sh
