out=gpurun_out/s10; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "p1 or P1 or fused or routes or tsx" > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
NOROT=$PWD/fem-elastoplasticity_amd/csrc/libfep_norot.so
for rep in 1 2 3; do for v in "A=1" "FEP_LIB_PATH=$NOROT"; do echo "== $v" >> $out/bench.log; env $v timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms'], 'kf', d['kf_only']['ms_per_step'], d['kf_only']['stream_ms_per_step'])" >> $out/bench.log; done; done; cat $out/bench.log
