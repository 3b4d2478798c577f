#!/bin/bash
out=gpurun_out/r4abl; mkdir -p $out
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
A=$PWD/fem-elastoplasticity_amd/csrc/libfep_hip_abl.so
FEP_LIB_PATH=$A timeout -k 10 600 python -m pytest tests/test_solver_gpu.py -x -q -m gpu -k "refresh_terms or bottom_of_the_cycle or block_transfers" > $out/pytest_abl_solver.log 2>&1; echo "ablation solver variant tests rc=$?"; tail -3 $out/pytest_abl_solver.log
FEP_LIB_PATH=$A timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu > $out/pytest_abl_parity.log 2>&1; echo "parity on the ablation build rc=$?"; tail -3 $out/pytest_abl_parity.log
for v in "FEP_P1_PATH=node_direct" "FEP_P1_PATH=node_list" "FEP_P1_PATH=node_unpacked" "FEP_P1_PATH=node2k" "FEP_P1_ASM=nodes" "FEP_P1_DMA=1" "FEP_P1_FUSED=all" "FEP_P1_TILE=128" "FEP_GEN_PATH=node" "FEP_PATCH_TPB=256" "FEP_NO_UNIFORM=1" "FEP_CSR_UNPACKED=1 FEP_ROUTE=coo"; do
  env FEP_LIB_PATH=$A $v timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "hot_path_vs_reference_golden or hot_path_mid_size or p1_fused_step" > $out/v.log 2>&1; echo "$v rc=$? $(tail -1 $out/v.log)"
done
