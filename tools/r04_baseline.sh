#!/bin/bash
# Round 4, first session: where the element route stands on THIS box + rocprof evidence for Q1 / Q2.
out=gpurun_out/r4a; mkdir -p $out
export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $out/rocm_smi_before.txt 2>&1
for t in "P2 708" "Q2 708" "Q1 708" "P4 354"; do echo "== $t" >> $out/elem_bench.log; python tools/elem_bench.py $t 30 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; done
echo "== P2 1414 random" >> $out/elem_bench.log; python tools/elem_bench.py P2 1414 10 random 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log
tools/prof.sh r04_q1 python3 tools/elem_bench.py Q1 708 10 > $out/prof_q1.log 2>&1
tools/prof.sh r04_q2 python3 tools/elem_bench.py Q2 708 10 > $out/prof_q2.log 2>&1
cat $out/elem_bench.log; tail -4 $out/prof_q1.log; tail -4 $out/prof_q2.log
