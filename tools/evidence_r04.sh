#!/bin/bash
# Round-4 evidence in ONE box session (box-to-box spread ~10 %: what is compared is measured together).
# Outputs under gpurun_out/ev4/ (copied into profiles/ afterwards).
out=gpurun_out/ev4; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
C=$PWD/fem-elastoplasticity_amd/csrc
rocm-smi --showclocks --showpower > $out/rocm_smi_before.txt 2>&1
if [ "$1" != "b" ]; then
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_driver_flags.json 2>> $out/bench_n1.err; echo "bench (driver flags) rc=$?"
for c in 1416 2832; do python bench.py --cells $c --steps 20 --no-cpu-baseline >> $out/bench_cells.jsonl 2>> $out/bench_cells.err; done
python bench.py --elem P2 --cells 708 --steps 20 > $out/bench_p2_708.json 2> $out/bench_p2.err; echo "bench P2 708 rc=$?"
python bench.py --elem P2 --state random --steps 10 --warmup 3 > $out/bench_p2_config5_n1.json 2>> $out/bench_p2.err; echo "bench configs[4] on one GPU rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 > $out/bench_2rank_gloo_weak.json 2> $out/bench_2rank.err; echo "2-rank weak rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong.json 2>> $out/bench_2rank.err; echo "2-rank strong rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --scaling strong --exchange p2p --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong_p2p.json 2>> $out/bench_2rank.err; echo "2-rank strong p2p rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --elem P2 --cells 708 --state random --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong_p2.json 2>> $out/bench_2rank.err; echo "2-rank strong P2 rc=$?"
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
python tools/setup_bench.py > $out/setup_bench.log 2>&1
python tools/host_path_bench.py > $out/host_path.log 2>&1
FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end.log 2>&1; echo "newton rc=$?"
python tools/newton_bench.py --inexact 1e-2 --cold > $out/newton_end_to_end_cold.log 2>&1; echo "newton (cold) rc=$?"
rocm-smi --showclocks --showpower > $out/rocm_smi_after.txt 2>&1
cut -c1-900 $out/bench_n1.json; tail -2 $out/bench_n1.err; cut -c1-700 $out/bench_2rank_gloo_strong.json; tail -1 $out/newton_end_to_end.log | cut -c1-300
else
tools/prof.sh r04_p1 --traffic-latest python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_p1.log 2>&1
FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_P1_TILE=128 tools/prof.sh r04_p1_tile128 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_p1_tile128.log 2>&1
tools/prof.sh r04_p2 python3 tools/elem_bench.py P2 708 10 > $out/prof_p2.log 2>&1
tools/prof.sh r04_p4 python3 tools/elem_bench.py P4 354 10 > $out/prof_p4.log 2>&1
tools/prof.sh r04_q1 python3 tools/elem_bench.py Q1 708 10 > $out/prof_q1.log 2>&1
tools/prof.sh r04_q2 python3 tools/elem_bench.py Q2 708 10 > $out/prof_q2.log 2>&1
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30" "P2 708 30 bands kf" "P2 1414 10 random"; do
run FEP_LIB_PATH=$C/libfep_hip_r03.so "$t"
run X=r04 "$t"
run FEP_ROUTE=coo "$t"
done
tail -4 $out/prof_p1.log; tail -3 $out/prof_p1_tile128.log; tail -3 $out/prof_p2.log; tail -3 $out/prof_p4.log; tail -3 $out/prof_q1.log; tail -3 $out/prof_q2.log; cut -c1-330 $out/elem_bench.log | grep -v "^Traceback\|^  "
fi
