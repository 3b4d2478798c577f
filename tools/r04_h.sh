#!/bin/bash
# P1: 128-block tiles on 128 threads (VERDICT r3 item 5's structural attempt) against the 256 default, same session
out=gpurun_out/r4h; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
T=$PWD/fem-elastoplasticity_amd/csrc/libfep_hip_abl_t128.so
FEP_LIB_PATH=$T FEP_P1_TILE=128 timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "p1_routes_agree or fused_step or full_size_p1 or unstructured or tsx_p1" > $out/pytest_t128.log 2>&1; echo "pytest t128 rc=$?"; tail -3 $out/pytest_t128.log
for i in 1 2 3; do
python bench.py --steps 50 --no-cpu-baseline > $out/bench_256_$i.json 2>> $out/bench.err
FEP_LIB_PATH=$T python bench.py --steps 50 --no-cpu-baseline > $out/bench_abl256_$i.json 2>> $out/bench.err
FEP_LIB_PATH=$T FEP_P1_TILE=128 FEP_VERBOSE=1 python bench.py --steps 50 --no-cpu-baseline > $out/bench_128_$i.json 2>> $out/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4h/bench_*.json')):
    try:
        j=json.load(open(f))
        print(f.split('/')[-1], 'step', round(j['ms_per_step'],4), 'frac', round(j['roofline']['frac'],3), 'kernels', {k:round(v,4) for k,v in j['roofline']['kernels_ms'].items()}, 'kf', round(j['kf_only']['ms_per_step'],4), round(j['kf_only']['frac'],3))
    except Exception as e:
        print(f, 'failed', e)
PY
grep "P1 plan" $out/bench.err | sort | uniq | head -3
