#!/usr/bin/env python3
"""cProfile of KrylovSolver.setup_amg on the strip-footing mesh (host part of the multigrid set-up)."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
fep = importlib.import_module('fem-elastoplasticity_amd')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = fep.square_mesh(N, 'P1', 10)
ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
ctx.set_materials(*bench.dp_materials())
K = ctx.step(np.zeros(ctx.n_dof), want=('K',))['K']
qf = mesh['Q'].flatten(order='F')
sol = fep.KrylovSolver(ctx, qf)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
sol.setup_amg(K.data, mesh['coordinates'])
pr.disable()
print('setup_amg', time.perf_counter() - t0, sol.amg_seconds)
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
