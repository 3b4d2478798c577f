out=gpurun_out/ev3d; mkdir -p $out
export TMPDIR=/tmp
FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end.log 2>&1; echo "newton rc=$?"
FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 --cold > $out/newton_end_to_end_cold.log 2>&1; echo "newton (cold) rc=$?"
FEP_AMG_REFRESH=0 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end_elastic_coarse.log 2>&1; echo "newton (elastic coarse operators) rc=$?"
FEP_AMG_FP32=0 FEP_AMG_BLOCK_TRANSFERS=0 FEP_AMG_PLAN=host FEP_PCG_FIXED_BATCH=1 FEP_AMG_TAIL=0 FEP_VERBOSE=1 python tools/newton_bench.py --inexact 1e-2 > $out/newton_end_to_end_switches_off.log 2>&1; echo "newton (switches off) rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nb -- python3 tools/newton_bench.py --inexact 1e-2 --steps 2 > $out/prof_newton.log 2>&1; find /tmp/prof_nb -name "*kernel_stats.csv" -exec cp {} $out/newton_kernel_stats.csv \;
for f in $out/newton_end_to_end*.log; do echo $f; grep "setup:\|set-up" $f | cut -c1-160; tail -1 $f | grep -o '"wall_s": [0-9.]*, "startup_s": [0-9.]*'; done
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
