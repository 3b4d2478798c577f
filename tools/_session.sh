out=gpurun_out/s16; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests/test_sharding_gpu.py tests/test_bench_contract_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "rc $?" >> $out/pytest.log; tail -4 $out/pytest.log
