#!/usr/bin/env python3
"""Offline study (CPU, SciPy) on tangents recorded by tools/capture_tangents.py: spectrum of the multigrid-preconditioned
operator seen by conjugate gradients (Lanczos coefficients of the run), and what deflating k recycled Ritz vectors of one
solve buys on the same and on the following tangents."""
import argparse
import importlib
import os
import sys

import numpy as np
import scipy.sparse as ssp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sol = importlib.import_module('fem-elastoplasticity_amd.solver')


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from deflation_study_lib import VCycle, pcg, lanczos_T  # noqa: E402


ap = argparse.ArgumentParser()
ap.add_argument('file')
ap.add_argument('--k', default='4,8,16,32')
a = ap.parse_args()
d = np.load(a.file)
ip, ix, free, xy = d['indptr'], d['indices'], d['free'].astype(bool), d['xy']
n = ip.size - 1
f = free.astype(np.float64)
Dq = ssp.diags(f)
Kref = ssp.csr_matrix((d['K_ref'], ix, ip), shape=(n, n))
levels = sol.build_amg_hierarchy(Kref, free, xy)
print('levels', [(n, Kref.nnz)] + [lv['size'] for lv in levels])
ids = sorted(int(k[1:]) for k in d.files if k[0] == 'K' and k[1:].isdigit())
sys_ = []
for i in ids:
    K = ssp.csr_matrix((d[f'K{i}'], ix, ip), shape=(n, n))
    A = (Dq @ K @ Dq + ssp.diags(1.0 - f)).tocsr()
    sys_.append((i, A, d[f'b{i}'] * f, int(d[f'it{i}'][0])))
Ael = (Dq @ Kref @ Dq + ssp.diags(1.0 - f)).tocsr()
M = VCycle(Ael, levels)
_, it_el, (al, be), _ = pcg(Ael, M, sys_[0][2], 1e-10)
ev = np.linalg.eigvalsh(lanczos_T(al, be))
print(f'elastic matrix: {it_el} iterations to 1e-10; Ritz values min {ev[0]:.4g} max {ev[-1]:.4g}; smallest 8: {np.round(ev[:8], 4)}')
Wprev = None
for i, A, b, it_gpu in sys_:
    M = VCycle(A, levels)
    M.A[1:] = [lv['A'] for lv in levels]
    x, it10, (al, be), Z = pcg(A, M, b, 1e-10, keep=400)
    _, it2, _, _ = pcg(A, M, b, 1e-2)
    T = lanczos_T(al, be)
    ev, Y = np.linalg.eigh(T)
    print(f'solve {i}: GPU run {it_gpu} its (1e-2); here {it2} its to 1e-2, {it10} to 1e-10; Ritz min {ev[0]:.4g} max {ev[-1]:.4g} cond {ev[-1] / ev[0]:.0f}')
    print('   smallest 24 Ritz values:', np.round(ev[:24], 4))
    m = min(len(Z), T.shape[0])
    Zm = np.stack(Z[:m], axis=1)
    Tm = T[:m, :m]
    evm, Ym = np.linalg.eigh(Tm)
    for k in [int(v) for v in a.k.split(',')]:
        W = Zm @ Ym[:, :k]
        _, d10, _, _ = pcg(A, M, b, 1e-10, W=W)
        _, d2, _, _ = pcg(A, M, b, 1e-2, W=W)
        line = f'   deflating its own {k} smallest Ritz vectors: {d2} its to 1e-2, {d10} to 1e-10'
        if Wprev is not None and k in Wprev:
            _, e10, _, _ = pcg(A, M, b, 1e-10, W=Wprev[k])
            _, e2, _, _ = pcg(A, M, b, 1e-2, W=Wprev[k])
            line += f' | recycled from the previous solve: {e2} / {e10}'
        print(line, flush=True)
    Wprev = {int(v): Zm @ Ym[:, :int(v)] for v in a.k.split(',')}

# ---- the production policy, replayed: every solve runs to 1e-2 with the current W, its k_new lowest Ritz vectors
# (from the Lanczos coefficients and the stored directions of that short run) go into W, oldest columns leave
for kmax, knew in ((32, 8), (32, 16), (16, 8), (48, 16)):
    W = None
    out = []
    for rep in range(2):
        for i, A, b, it_gpu in sys_:
            M = VCycle(A, levels)
            M.A[1:] = [lv['A'] for lv in levels]
            _, it, (al, be), Z = pcg(A, M, b, 1e-2, W=W, keep=400)
            out.append((i, it_gpu, it))
            m = min(len(Z), len(al))
            if m >= 2:
                ev, Y = np.linalg.eigh(lanczos_T(al[:m], be[:m - 1]))
                new = np.stack(Z[:m], axis=1) @ Y[:, :min(knew, m)]
                W = new if W is None else np.concatenate([W, new], axis=1)[:, -kmax:]
    print(f'policy kmax {kmax} knew {knew}: (solve, undeflated, deflated) {out}', flush=True)
