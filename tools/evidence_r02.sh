#!/bin/bash
# Round-2 evidence in ONE box session: bench lines, scaling rehearsal, rocprofv3 passes.  Outputs under gpurun_out/ev/
# (copied into profiles/ afterwards).
out=gpurun_out/ev; mkdir -p $out
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
for c in 1416 2832; do python bench.py --cells $c --steps 20 --no-cpu-baseline >> $out/bench_cells.jsonl 2>> $out/bench_cells.err; done
python bench.py --state newton --no-cpu-baseline > $out/bench_newton_state.json 2>> $out/bench_cells.err
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 > $out/bench_2rank_gloo_weak.json 2> $out/bench_2rank.err; echo "2-rank weak rc=$?"
FEP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --backend gloo --scaling strong --steps 10 --warmup 2 > $out/bench_2rank_gloo_strong.json 2>> $out/bench_2rank.err; echo "2-rank strong rc=$?"
tools/prof.sh r02_p1 --traffic-latest python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_p1.log 2>&1
tools/prof.sh r02_p1_2832 python3 bench.py --cells 2832 --steps 5 --warmup 2 --no-cpu-baseline > $out/prof_p1_2832.log 2>&1
tools/prof.sh r02_p2 python3 tools/elem_bench.py P2 708 10 > $out/prof_p2.log 2>&1
python tools/setup_bench.py > $out/setup_bench.log 2>&1
for t in P2 Q2 Q1; do python tools/elem_bench.py $t 708 20 >> $out/elem_bench.log 2>&1; done
python tools/elem_bench.py P2 1414 10 random >> $out/elem_bench.log 2>&1
tail -4 $out/prof_p1.log; tail -3 $out/prof_p1_2832.log; tail -4 $out/prof_p2.log; cat $out/bench_n1.json | cut -c1-600
