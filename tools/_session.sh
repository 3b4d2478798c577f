out=gpurun_out/s12; mkdir -p $out
export TMPDIR=/tmp
for rep in 1 2; do for v in "A=1" "FEP_LIB_PATH=$PWD/fem-elastoplasticity_amd/csrc/libfep_passes2.so" "FEP_LIB_PATH=$PWD/fem-elastoplasticity_amd/csrc/libfep_passes1.so"; do echo "== $v" >> $out/newton_ab.log; env $v timeout -k 10 300 python tools/newton_bench.py --inexact 1e-2 2>&1 | grep "setup:\|wall_s" | cut -c1-300 | sed 's/"newton_its.*"wall_s"/"wall_s"/' >> $out/newton_ab.log; done; done; cat $out/newton_ab.log
