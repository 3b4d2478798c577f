#!/bin/bash
out=gpurun_out/r4b; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/pytest_parity.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_parity.log
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
run X=1 "P2 708 30"
run X=1 "Q2 708 30"
run FEP_PATCH_TPB=384 FEP_PATCH_JS=2 "Q2 708 30"
run X=1 "Q1 708 30"
run X=1 "P4 354 30"
run FEP_PATCH_TPB=256 FEP_PATCH_JS=1 "P4 354 30"
done
run X=1 "P2 1414 10 random"
run FEP_PATCH_DBG=16 "P2 708 30"
run FEP_PATCH_DBG=8 "P2 708 30"
cat $out/elem_bench.log
