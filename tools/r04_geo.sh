#!/bin/bash
# geometry mode of the element kernel per type on the final kernel (ablation build: FEP_ELEM_GEO=0|1)
out=gpurun_out/r4geo; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
A=$PWD/fem-elastoplasticity_amd/csrc/libfep_hip_abl.so
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P4 354 30" "Q2 708 30" "P2 708 30" "Q1 708 30"; do
run FEP_LIB_PATH=$A FEP_ELEM_GEO=0 "$t"
run FEP_LIB_PATH=$A FEP_ELEM_GEO=1 "$t"
done
done
FEP_LIB_PATH=$A FEP_ELEM_GEO=1 timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "p4 or P4" > $out/pytest_p4geo.log 2>&1; echo "P4 GEO parity rc=$?"; tail -2 $out/pytest_p4geo.log
cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  "
