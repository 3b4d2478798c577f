#!/usr/bin/env python3
"""End-to-end run of BASELINE configs[3] on one GPU: strip footing, N x N cells of P1 elements, `--steps` accepted
load steps, Newton iterate resident on the device (linear_solver='pcg').  Prints wall time, the share of the hot
path, and the PCG iteration counts.  Not the bench metric (bench.py times the hot path alone)."""
import argparse
import importlib
import json
import os
import sys
import time

t_proc = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fep = importlib.import_module('fem-elastoplasticity_amd')

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=708)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--element', default='P1')
ap.add_argument('--rtol', type=float, default=1e-10)
ap.add_argument('--forcing', type=float, default=0.0, help='inexact Newton: linear rtol = forcing * previous Newton criterion (0 = fixed rtol)')
ap.add_argument('--inexact', type=float, default=0.0, help='constant relative tolerance of every linear solve inside the Newton loop (0 = off)')
ap.add_argument('--cap', type=float, default=1e-4, help='loosest linear tolerance the forcing term may ask for')
ap.add_argument('--solver', default='amg', help='pcg (block-Jacobi CG) | amg (multigrid-preconditioned CG) | direct')

ap.add_argument('--cold', action='store_true', help='leave the library load (with `import torch`) and the HIP runtime start inside the timed call, as the runs before round 3\'s last session did')
a = ap.parse_args()
if not a.cold:
    # what a process pays once, whatever it goes on to compute: loading the library (which imports torch first, for its HIP
    # runtime) and starting the HIP runtime on the device.  Reported as `startup_s`, not part of `wall_s`.
    import torch
    fep.lib()
    torch.zeros(1, device='cuda')
    torch.cuda.synchronize()
t_start = time.perf_counter() - t_proc

lines = []
t0 = time.perf_counter()
h = fep.solve_strip_footing(a.element, n_cells=a.n, max_steps=a.steps, linear_solver=a.solver, pcg_rtol=a.rtol, pcg_forcing=a.forcing or None, pcg_forcing_cap=a.cap, pcg_inexact_rtol=a.inexact or None,
                            keep_U=False, log=lambda s: (lines.append(s), print(f'[{time.perf_counter() - t0:8.2f}s] {s}', flush=True)))
t = time.perf_counter() - t0
it = h['pcg_iters'] or []
print(json.dumps({'n_cells': a.n, 'element': a.element, 'elements': int(h['mesh']['elements'].shape[1]),
                  'accepted_steps': len(h['zeta']), 'hot_path_calls': h['n_calls'], 'newton_its': h['newton_its'],
                  'wall_s': t, 'startup_s': t_start, 'cold': bool(a.cold), 'linear_solves': len(it), 'pcg_iters_total': int(sum(it)),
                  'pcg_iters_max': int(max(it)) if it else None, 'pcg_iters': [int(v) for v in it], 'zeta': h['zeta'], 'pressure': h['pressure'],
                  'counts': h['counts']}))
