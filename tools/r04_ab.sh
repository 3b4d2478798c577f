#!/bin/bash
# In-session A/B of kernel versions (round 4's way of measuring: boxes of the pool differ by +-10 %, runs on one box by +-5 %).
# Every library to compare sits in csrc/ next to the product (built from another checkout / with other -D flags; *.so travel
# to the GPU box, they are git-ignored) and is selected with FEP_LIB_PATH; variants are interleaved, two passes.
#   csrc/libfep_hip_r03.so   round 3's kernels: `git show 6f91ac8:fem-elastoplasticity_amd/csrc/<file>` into a scratch
#                            directory, the entry points added since (fep_iface_sum_f64, fep_build_is_ablation,
#                            fep_ctx_kernel_names) stubbed in, compiled with build.py's command line
#   csrc/libfep_hip_abl.so   `python fem-elastoplasticity_amd/build.py --ablation`: FEP_PHASE_CLK=1 prints the mean shader
#                            clocks per phase of a workgroup of element_kernel when the context is destroyed
# usage: gpurun -- 'bash tools/r04_ab.sh [more libfep_hip_<tag>.so tags]'
out=gpurun_out/r4ab; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
C=$PWD/fem-elastoplasticity_amd/csrc
run() { echo "== $*" >> $out/elem_bench.log; env "${@:1:$#-1}" python tools/elem_bench.py ${!#} 2>&1 | grep -v amdgpu.ids >> $out/elem_bench.log; }
for i in 1 2; do
for t in "P2 708 30" "Q2 708 30" "Q1 708 30" "P4 354 30"; do
[ -f $C/libfep_hip_r03.so ] && run FEP_LIB_PATH=$C/libfep_hip_r03.so "$t"
run X=product "$t"
for tag in "$@"; do run FEP_LIB_PATH=$C/libfep_hip_$tag.so "$t"; done
done
done
for t in "P2 708 5" "Q2 708 5" "Q1 708 5" "P4 354 5"; do
run FEP_LIB_PATH=$C/libfep_hip_abl.so FEP_PHASE_CLK=1 "$t"
done
cut -c1-400 $out/elem_bench.log | grep -v "^Traceback\|^  "
