out=gpurun_out/s5; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_solver_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -5 $out/pytest.log
for v in "A=1" "FEP_AMG_BLOCK_TRANSFERS=0" "A=1" "FEP_AMG_BLOCK_TRANSFERS=0"; do echo "== $v" >> $out/newton_ab.log; env $v FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 2>&1 | grep "setup:\|set-up\|wall_s" | cut -c1-300 | sed 's/"newton_its.*"wall_s"/"wall_s"/' >> $out/newton_ab.log; done; cat $out/newton_ab.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nb -- python3 tools/newton_bench.py --inexact 1e-2 --steps 2 > $out/prof_newton.log 2>&1; find /tmp/prof_nb -name "*kernel_stats.csv" -exec cp {} $out/newton_kernel_stats.csv \; ; head -12 $out/newton_kernel_stats.csv | cut -c1-200
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; cut -c1-1500 $out/bench.json
