#!/bin/bash
# Round-3 evidence in ONE box session (box-to-box spread ~10 %: what is compared is measured together).
# Outputs under gpurun_out/ev3/ (copied into profiles/ afterwards).
out=gpurun_out/ev3; mkdir -p $out
rocm-smi --showclocks --showpower > $out/rocm_smi_before.txt 2>&1
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
for c in 1416 2832; do python bench.py --cells $c --steps 20 --no-cpu-baseline >> $out/bench_cells.jsonl 2>> $out/bench_cells.err; done
python bench.py --state newton --no-cpu-baseline > $out/bench_newton_state.json 2>> $out/bench_cells.err
python bench.py --elem P2 --cells 708 --steps 20 > $out/bench_p2_708.json 2> $out/bench_p2.err; echo "bench P2 708 rc=$?"
python bench.py --elem P2 --state random --steps 10 --warmup 3 > $out/bench_p2_config5_n1.json 2>> $out/bench_p2.err; echo "bench configs[4] on one GPU rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 > $out/bench_2rank_gloo_weak.json 2> $out/bench_2rank.err; echo "2-rank weak rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong.json 2>> $out/bench_2rank.err; echo "2-rank strong rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --elem P2 --cells 708 --state random --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong_p2.json 2>> $out/bench_2rank.err; echo "2-rank strong P2 rc=$?"
tools/prof.sh r03_p1 --traffic-latest python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_p1.log 2>&1
tools/prof.sh r03_p2 python3 tools/elem_bench.py P2 708 10 > $out/prof_p2.log 2>&1
tools/prof.sh r03_p4 python3 tools/elem_bench.py P4 354 10 > $out/prof_p4.log 2>&1
FEP_GEN_PATH=coo tools/prof.sh r03_p2_coo python3 tools/elem_bench.py P2 708 10 > $out/prof_p2_coo.log 2>&1
for t in "P2 708" "Q2 708" "Q1 708" "P4 354"; do for v in "FEP_GEN_PATH=patch" "FEP_GEN_PATH=coo"; do echo "== $t $v" >> $out/elem_bench.log; env $v python tools/elem_bench.py $t 30 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; done; done
for v in "FEP_GEN_PATH=patch" "FEP_GEN_PATH=coo"; do echo "== P2 708 K,F-only $v" >> $out/elem_bench.log; env $v python tools/elem_bench.py P2 708 30 bands kf 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; echo "== P2 1414 random $v" >> $out/elem_bench.log; env $v python tools/elem_bench.py P2 1414 10 random 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; done
python tools/setup_bench.py > $out/setup_bench.log 2>&1
python tools/host_path_bench.py > $out/host_path.log 2>&1
for v in "FEP_PATCH_TPB=256"; do echo "== P2 708 $v" >> $out/elem_bench.log; env $v python tools/elem_bench.py P2 708 30 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; echo "== P2 1414 random $v" >> $out/elem_bench.log; env $v python tools/elem_bench.py P2 1414 10 random 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; done
FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end.log 2>&1; echo "newton rc=$?"
FEP_AMG_REFRESH=0 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end_elastic_coarse.log 2>&1; echo "newton (elastic coarse operators) rc=$?"
TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nb -- python3 tools/newton_bench.py --inexact 1e-2 --steps 2 > $out/prof_newton.log 2>&1; find /tmp/prof_nb -name "*kernel_stats.csv" -exec cp {} $out/newton_kernel_stats.csv \;
rocm-smi --showclocks --showpower > $out/rocm_smi_after.txt 2>&1
tail -4 $out/prof_p1.log; tail -4 $out/prof_p2.log; tail -3 $out/prof_p4.log; cut -c1-700 $out/bench_n1.json; cat $out/bench_n1.err | tail -2; tail -1 $out/newton_end_to_end.log | cut -c1-300
