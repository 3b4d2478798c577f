mkdir -p gpurun_out/r15
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_solver_gpu.py -x -q -m gpu > gpurun_out/r15/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r15/pytest.log; tail -4 gpurun_out/r15/pytest.log
FEP_VERBOSE=1 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 > gpurun_out/r15/newton_refresh.log 2>&1; grep "setup\|set-up" gpurun_out/r15/newton_refresh.log; tail -1 gpurun_out/r15/newton_refresh.log | cut -c1-300
FEP_AMG_FP32=0 timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 > gpurun_out/r15/newton_fp64.log 2>&1; tail -1 gpurun_out/r15/newton_fp64.log | cut -c1-300
