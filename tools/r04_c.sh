#!/bin/bash
# in-session A/B: round-3 kernels (libfep_hip_r03.so) against the current ones, interleaved; phase clocks of the ablation build
out=gpurun_out/r4c; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
R03=$PWD/fem-elastoplasticity_amd/csrc/libfep_hip_r03.so
ABL=$PWD/fem-elastoplasticity_amd/csrc/libfep_hip_abl.so
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run FEP_LIB_PATH=$R03 "$t"
run X=new "$t"
done
run FEP_PATCH_TPB=256 FEP_PATCH_JS=1 "P4 354 30"
done
run FEP_LIB_PATH=$R03 "P2 1414 10 random"
run X=new "P2 1414 10 random"
for t in "P2 708 5" "Q2 708 5" "Q1 708 5" "P4 354 5"; do
run FEP_LIB_PATH=$ABL FEP_PHASE_CLK=1 "$t"
done
run FEP_LIB_PATH=$ABL FEP_PHASE_CLK=1 FEP_PATCH_TPB=256 FEP_PATCH_JS=1 "P4 354 5"
run FEP_LIB_PATH=$ABL FEP_PHASE_CLK=1 FEP_PATCH_TPB=256 FEP_PATCH_JS=1 "P2 708 5"
cat $out/elem_bench.log | cut -c1-400
