#!/usr/bin/env python3
"""Writes tests/golden/newton_354_pins.json: load history and footing pressures of BASELINE configs[3]'s driver at 354 x 354
cells (250 632 P1 elements, 10 load steps) as the multigrid-CG solver with two-digit linear solves gives them
(`tools/newton_bench.py --n 354 --inexact 1e-2`), cross-checked in the same run against ten-digit linear solves — the
regression guard of tests/test_solver_gpu.py.  Needs a GPU."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fep = importlib.import_module('fem-elastoplasticity_amd')


def run(inexact):
    h = fep.solve_strip_footing('P1', n_cells=354, max_steps=10, linear_solver='amg', pcg_rtol=1e-10,
                                pcg_inexact_rtol=inexact, keep_U=False)
    it = h['pcg_iters'] or []
    return {'zeta': [float(z) for z in h['zeta']], 'pressure': [float(p) for p in h['pressure']], 'hot_path_calls': int(h['n_calls']),
            'newton_its': [int(v) for v in h['newton_its']], 'pcg_iters_total': int(sum(it)), 'pcg_iters_max': int(max(it)),
            'counts': [[int(a), int(b)] for a, b in h['counts']] if h.get('counts') is not None else None}


a = run(1e-2)
b = run(None)
dev = max(abs(x - y) / abs(y) for x, y in zip(a['pressure'], b['pressure']))
assert a['zeta'] == b['zeta'], (a['zeta'], b['zeta'])
out = {'n_cells': 354, 'elements': 250632, 'inexact_1e-2': a, 'rtol_1e-10': b, 'max_rel_pressure_difference': dev,
       'note': 'P1 strip footing, 10 load steps; pressures = normalised footing pressure logged at the next step (DP:1032-1034)'}
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'tests', 'golden', 'newton_354_pins.json')
json.dump(out, open(path, 'w'), indent=1)
print(json.dumps(out))
