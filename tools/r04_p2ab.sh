#!/bin/bash
out=gpurun_out/r4p2; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
C=$PWD/fem-elastoplasticity_amd/csrc
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2 3 4 5; do
for t in "P2 708 30" "P2 1414 10 random" "P4 354 30"; do
run FEP_LIB_PATH=$C/libfep_hip_r03.so "$t"
run X=r04 "$t"
run FEP_LIB_PATH=$C/libfep_hip_nh.so "$t"
done
done
python - <<'PY'
import re, collections
txt=open('gpurun_out/r4p2/elem_bench.log').read().split('\n')
res=collections.defaultdict(list)
for l in txt:
    if l.startswith('=='):
        cur=('r03' if 'r03' in l else 'nohoist' if '_nh' in l else 'r04', re.search(r'(P\d) (\d+)',l).group(0))
    elif 'step' in l and 'ms ->' in l:
        m=re.search(r'step ([\d.]+) ms.*\'element\': ([\d.]+), \'csr\': ([\d.]+)',l)
        res[cur].append(float(m.group(1)))
for k in sorted(res): print(k, ' '.join(f'{v:.3f}' for v in res[k]), ' median', sorted(res[k])[len(res[k])//2])
PY
