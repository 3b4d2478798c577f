out=gpurun_out/s14; mkdir -p $out
export TMPDIR=/tmp
for n in 354 1001 1416; do echo "== n $n" >> $out/newton_sizes.log; FEP_VERBOSE=1 timeout -k 10 500 python tools/newton_bench.py --inexact 1e-2 --n $n 2>&1 | grep "setup:\|set-up\|wall_s" | cut -c1-330 | sed 's/"newton_its.*"wall_s"/"wall_s"/' >> $out/newton_sizes.log; done; cat $out/newton_sizes.log
