#!/bin/bash
# Two patches per workgroup (ablation build, FEP_NPATCH=2): the second patch's ids fetched with the first one's phase-3 descriptors,
# its node data by LDS-DMA under the first one's phase 3.  Parity under the switch, then in-session A/B (SHA-1 of K and F).
out=gpurun_out/r4npatch; mkdir -p $out; rm -f $out/*.log
export TMPDIR=/tmp
C=$PWD/fem-elastoplasticity_amd/csrc
FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_NPATCH=2 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/parity.log 2>&1; rc=$?; tail -3 $out/parity.log
[ $rc -ne 0 ] && exit $rc
export FEP_BENCH_HASH=1
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2 3; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30"; do
run FEP_LIB_PATH=$C/libfep_hip_abl.so "$t"
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_NPATCH=2 "$t"
done
done
run FEP_LIB_PATH=$C/libfep_hip_abl.so "P2 1414 10 random"
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_NPATCH=2 "P2 1414 10 random"
for t in "P2 708 5" "Q2 708 5"; do run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_NPATCH=2 FEP_PHASE_CLK=1 "$t"; done
python3 - $out/elem_bench.log <<'PY'
import re,sys
cur=None
for l in open(sys.argv[1]):
    if l.startswith('=='): cur=re.sub(r'FEP_LIB_PATH=\S*libfep_hip_','',l[3:].strip())
    elif re.match(r'^[PQ][124]',l):
        m=re.search(r"step ([0-9.]+) ms.*'element': ([0-9.]+), 'csr': ([0-9.]+)",l); print(f"{cur:48s} step {m.group(1)} element {m.group(2)} fixup {m.group(3)}", end=' ')
    elif 'sha1' in l: print(l.strip())
    elif 'phase clocks' in l or 'staging of' in l: print(l.strip()[:260])
PY
