#!/bin/bash
out=gpurun_out/r4clk; mkdir -p $out; rm -f $out/elem_bench.log
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
C=$PWD/fem-elastoplasticity_amd/csrc
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for t in "P2 708 30 bands" "P2 1414 10 bands" "P2 708 30 bands" "P2 1414 10 random" "P4 354 30" "Q2 708 30"; do run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_PHASE_CLK=1 "$t"; done
grep -v "^Traceback\|^  " $out/elem_bench.log | cut -c1-420
