mkdir -p gpurun_out/r8
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_sharding_gpu.py tests/test_bench_contract_gpu.py -x -q -m gpu > gpurun_out/r8/pytest.log 2>&1; echo "rc $?" >> gpurun_out/r8/pytest.log; tail -3 gpurun_out/r8/pytest.log
tools/r03_matrix.sh gpurun_out/r8 "Q2 708" FEP_VERBOSE=1 FEP_PATCH_TPB=512,FEP_VERBOSE=1 FEP_PATCH_TPB=512,FEP_PATCH_RUNS=4,FEP_VERBOSE=1
tools/r03_matrix.sh gpurun_out/r8 "P2 708" FEP_VERBOSE=1 FEP_PATCH_TPB=256
tools/r03_matrix.sh gpurun_out/r8 "P2 1414 10 random" FEP_VERBOSE=0 FEP_PATCH_TPB=256
grep -B1 "step" gpurun_out/r8/matrix.log | grep -v "^--" | cut -c1-175
