out=gpurun_out/s8; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_solver_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
for v in "A=1" "FEP_AMG_TAIL=0" "A=1" "FEP_AMG_TAIL=0"; do echo "== $v" >> $out/newton_ab.log; env $v FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 2>&1 | grep "setup:\|wall_s" | cut -c1-300 | sed 's/"newton_its.*"wall_s"/"wall_s"/' >> $out/newton_ab.log; done; cat $out/newton_ab.log
