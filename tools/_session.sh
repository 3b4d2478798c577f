out=gpurun_out/s13; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_solver_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
for v in "A=1" "A=1"; do echo "== $v" >> $out/newton_ab.log; env $v FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 2>&1 | grep "setup:\|wall_s" | cut -c1-300 | sed 's/"newton_its.*"wall_s"/"wall_s"/' >> $out/newton_ab.log; done; cat $out/newton_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nb -- python3 tools/newton_bench.py --inexact 1e-2 --steps 2 > $out/prof_newton.log 2>&1; find /tmp/prof_nb -name "*kernel_stats.csv" -exec cp {} $out/newton_kernel_stats.csv \; ; grep "block_residual\|to_float" $out/newton_kernel_stats.csv | cut -c1-70,330-420
