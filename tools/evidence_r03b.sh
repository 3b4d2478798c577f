#!/bin/bash
# Round-3 evidence, last session of the round (after the solver / set-up work and bench.py's pre-heat + stream-event roofline).
# The element-route kernels are those of tools/evidence_r03.sh: its P2 / P4 counter passes and the elem_bench matrix stand.
# Outputs under gpurun_out/ev3b/ (copied into profiles/ afterwards).
out=gpurun_out/ev3b; mkdir -p $out
export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $out/rocm_smi_before.txt 2>&1
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $out/bench_n1_driver_flags.json 2>> $out/bench_n1.err; echo "bench (driver's flags) rc=$?"
python bench.py --steps 20 --warmup 5 --preheat-ms 0 --no-cpu-baseline > $out/bench_n1_no_preheat.json 2>> $out/bench_n1.err
for c in 1416 2832; do python bench.py --cells $c --steps 20 --no-cpu-baseline >> $out/bench_cells.jsonl 2>> $out/bench_cells.err; done
python bench.py --state newton --no-cpu-baseline > $out/bench_newton_state.json 2>> $out/bench_cells.err
python bench.py --elem P2 --cells 708 --steps 20 > $out/bench_p2_708.json 2> $out/bench_p2.err; echo "bench P2 708 rc=$?"
python bench.py --elem P2 --state random --steps 10 --warmup 3 > $out/bench_p2_config5_n1.json 2>> $out/bench_p2.err; echo "bench configs[4] on one GPU rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 > $out/bench_2rank_gloo_weak.json 2> $out/bench_2rank.err; echo "2-rank weak rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong.json 2>> $out/bench_2rank.err; echo "2-rank strong rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --elem P2 --cells 708 --state random --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong_p2.json 2>> $out/bench_2rank.err; echo "2-rank strong P2 rc=$?"
tools/prof.sh r03_p1 --traffic-latest python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_p1.log 2>&1; tail -3 $out/prof_p1.log
FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end.log 2>&1; echo "newton rc=$?"
FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 --cold > $out/newton_end_to_end_cold.log 2>&1; echo "newton (cold) rc=$?"
FEP_AMG_REFRESH=0 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end_elastic_coarse.log 2>&1; echo "newton (elastic coarse operators) rc=$?"
FEP_AMG_FP32=0 FEP_AMG_BLOCK_TRANSFERS=0 FEP_AMG_PLAN=host FEP_PCG_FIXED_BATCH=1 FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end_switches_off.log 2>&1; echo "newton (this session's changes switched off) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nb -- python3 tools/newton_bench.py --inexact 1e-2 --steps 2 > $out/prof_newton.log 2>&1; find /tmp/prof_nb -name "*kernel_stats.csv" -exec cp {} $out/newton_kernel_stats.csv \;
python tools/setup_bench.py > $out/setup_bench.log 2>&1
rocm-smi --showclocks --showpower > $out/rocm_smi_after.txt 2>&1
cut -c1-600 $out/bench_n1.json; tail -2 $out/bench_n1.err; for f in $out/newton_end_to_end*.log; do echo $f; grep "setup:" $f | cut -c1-160; tail -1 $f | cut -c1-120 | sed 's/"newton_its.*//'; tail -1 $f | grep -o '"wall_s": [0-9.]*, "startup_s": [0-9.]*'; done
