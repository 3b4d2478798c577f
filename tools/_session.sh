out=gpurun_out/s7; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_solver_gpu.py tests/test_newton_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
for v in "A=1" "A=1"; do echo "== $v" >> $out/newton_ab.log; env $v FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 2>&1 | grep "setup:\|set-up\|wall_s" | cut -c1-300 | sed 's/"newton_its.*"wall_s"/"wall_s"/' >> $out/newton_ab.log; done; cat $out/newton_ab.log
timeout -k 10 200 python tools/amg_setup_profile.py > $out/amg_profile.log 2>&1; grep -v "^$" $out/amg_profile.log | head -32
