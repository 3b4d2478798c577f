out=gpurun_out/ev3e; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-400 $out/bench_driver_flags.json
