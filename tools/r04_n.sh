#!/bin/bash
out=gpurun_out/ev4; mkdir -p $out
export TMPDIR=/tmp
python -c "import importlib,sys; sys.path.insert(0,'.'); print(importlib.import_module('fem-elastoplasticity_amd').build())"
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest (all gpu) rc=$?"; tail -4 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/evidence_r04.sh a
