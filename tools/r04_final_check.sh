#!/bin/bash
out=gpurun_out/r4fin2; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest (all gpu) rc=$?"; tail -4 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
C=$PWD/fem-elastoplasticity_amd/csrc
for t in "p2 P2 708" "p4 P4 354" "q1 Q1 708" "q2 Q2 708"; do set -- $t; tools/prof.sh r04_$1 python3 tools/elem_bench.py $2 $3 10 > $out/prof_$1.log 2>&1; done
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2 3; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
run FEP_LIB_PATH=$C/libfep_hip_r03.so "$t"
run X=r04 "$t"
done
done
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_PHASE_CLK=1 "P4 354 5"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench_driver_flags.json
tail -3 $out/prof_p4.log; cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  "
