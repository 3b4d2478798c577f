#!/usr/bin/env python3
"""Records the linear systems of a few consecutive Newton iterates of the strip-footing run (K values on the CSR
pattern, right-hand side, free-DOF mask, node coordinates) for offline studies of the solver's convergence
(tools/deflation_study.py).  `--solves a,b,c`: 0-based indices of the linear solves to keep."""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fep = importlib.import_module('fem-elastoplasticity_amd')
solver_mod = importlib.import_module('fem-elastoplasticity_amd.solver')

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=256)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--solves', default='40,41,42')
ap.add_argument('--out', default='gpurun_out/tangents.npz')
a = ap.parse_args()
keep = [int(v) for v in a.solves.split(',')]
rec, state = {}, {'i': 0}

orig_pcg = solver_mod.KrylovSolver.pcg
orig_amg = solver_mod.KrylovSolver.setup_amg


def pcg(self, k_data, b, **kw):
    i = state['i']
    state['i'] += 1
    if i in keep:
        rec[f'K{i}'] = k_data.detach().cpu().numpy().copy()
        rec[f'b{i}'] = b.detach().cpu().numpy().copy()
    x = orig_pcg(self, k_data, b, **kw)
    if i in keep:
        rec[f'it{i}'] = np.array([self.last['iters']])
        ip, ix = self._pattern
        rec['indptr'], rec['indices'], rec['free'] = ip, ix, self.free_dof.copy()
    return x


def setup_amg(self, K_ref, coordinates, **kw):
    rec['K_ref'] = np.asarray(K_ref.data if hasattr(K_ref, 'indptr') else K_ref).copy()
    rec['xy'] = np.asarray(coordinates).copy()
    return orig_amg(self, K_ref, coordinates, **kw)


solver_mod.KrylovSolver.pcg = pcg
solver_mod.KrylovSolver.setup_amg = setup_amg
h = fep.solve_strip_footing('P1', n_cells=a.n, max_steps=a.steps, linear_solver='amg', pcg_rtol=1e-10, pcg_inexact_rtol=1e-2,
                            keep_U=False, log=lambda s: print(s, flush=True))
print('solves', state['i'], 'iters', h['pcg_iters'])
np.savez_compressed(a.out, **rec)
print('wrote', a.out, {k: v.shape for k, v in rec.items()})
