#!/bin/bash
# BASELINE configs[3] end to end (tools/newton_bench.py, 708 x 708 cells, 10 load steps, multigrid CG) for several
# inexact-Newton forcing terms: `tools/newton_forcing.sh OUT "0.01 0.1 0.3" [extra newton_bench args]`
out=${1:-gpurun_out/newton_forcing.log}; shift
fs=${1:-0.01 0.1}; shift
for f in $fs; do
  python tools/newton_bench.py --forcing $f "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('forcing $f: wall %.1f s, accepted %d, hot-path calls %d, solves %d, pcg iterations %d (max %d), newton its %s, last pressure %.8f'
      % (j['wall_s'], j['accepted_steps'], j['hot_path_calls'], j['linear_solves'], j['pcg_iters_total'], j['pcg_iters_max'],
         j['newton_its'], j['pressure'][-1]))" | tee -a $out
done
