#!/usr/bin/env python3
"""Multigrid-preconditioned CG on K_elast of the strip-footing mesh: hierarchy sizes, setup time, iterations, time per solve.
    python tools/amg_bench.py [N ...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
coarse = int(os.environ.get('AMG_COARSE_NODES', '400'))
for N in [int(a) for a in sys.argv[1:]] or [256]:
    mesh = fep.square_mesh(N, 'P1', 10)
    ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
    ctx.set_materials(*bench.dp_materials())
    K = ctx.step(np.zeros(ctx.n_dof), want=('K',))['K']
    qf = mesh['Q'].flatten(order='F')
    sol = fep.KrylovSolver(ctx, qf)
    t0 = time.perf_counter()
    levels = sol.setup_amg(K, mesh['coordinates'], coarse_nodes=coarse)
    t_setup = time.perf_counter() - t0
    dev = torch.device('cuda', 0)
    kd = torch.from_numpy(K.data).to(dev)
    b = torch.from_numpy(np.random.default_rng(0).normal(size=ctx.n_dof)).to(dev)
    for pre in ('amg', 'jacobi'):
        sol.pcg(kd, b, rtol=1e-10, precond=pre, max_iter=200 if pre == 'amg' else 100)      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = sol.pcg(kd, b, rtol=1e-10, precond=pre)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f'N={N} dofs={ctx.n_dof} {pre}: {sol.last["iters"]} iterations, {t*1e3:.1f} ms, '
              f'{t/max(sol.last["iters"],1)*1e6:.0f} us/iteration, state {sol.last["state"]}', flush=True)
    print(f'N={N} levels {levels}, setup {t_setup:.1f} s', flush=True)
    sol.close(); ctx.close()
