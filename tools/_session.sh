out=gpurun_out/s1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_solver_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 200 --warmup 50" "--steps 20 --warmup 5" "--steps 2000 --warmup 500"; do echo "== $a" >> $out/bench.log; timeout -k 10 200 python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms'], d['kf_only']['ms_per_step'])" >> $out/bench.log; done; cat $out/bench.log
FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 > $out/newton.log 2>&1; grep "setup\|set-up" $out/newton.log | cut -c1-200; tail -1 $out/newton.log | cut -c1-330
FEP_AMG_FP32=1 FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 > $out/newton_fp32.log 2>&1; grep "setup\|set-up" $out/newton_fp32.log | cut -c1-200; tail -1 $out/newton_fp32.log | cut -c1-330
