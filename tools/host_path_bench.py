#!/usr/bin/env python3
"""Timings of the host-array entry points (what the reference's drivers call): the drop-in return map on n points and
the fused step on the 708x708 P1 mesh, PCIe transfers included."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
n = 1002528
rng = np.random.default_rng(2024)
e = np.asfortranarray(rng.normal(0, 2e-4, size=(3, n)))
ep = np.zeros((4, n))
sh, bu, eta, c = [np.full(n, v) for v in bench.dp_materials()]
cc = fep.plasticity2d_dp.construct_constitutive_problem
cc(e, ep, sh, bu, eta, c)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = cc(e, ep, sh, bu, eta, c); ts.append(time.perf_counter() - t0)
mb = (e.nbytes + ep.nbytes + 4 * sh.nbytes + r['s'].nbytes + r['ds'].nbytes + n) / 1e6
print(f'construct_constitutive_problem, {n} points: {min(ts)*1e3:.1f} ms ({n/min(ts)/1e6:.1f} M points/s), {mb:.0f} MB over PCIe '
      f'-> {mb/1e3/min(ts):.1f} GB/s; plastic points {int(r["ind_p"].sum())}')
mesh = fep.square_mesh(708, 'P1', 10)
ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
ctx.set_materials(*bench.dp_materials())
U = bench.displacement(mesh['coordinates'])
Ep = np.zeros((4, ctx.n_int))
for want in (('K', 'F'), ('s', 'ds', 'ind_p', 'K', 'F')):
    ctx.step(U, Ep, want=want)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = ctx.step(U, Ep, want=want); ts.append(time.perf_counter() - t0)
    print(f'MeshContext.step want={want}: {min(ts)*1e3:.1f} ms ({ctx.n_int/min(ts)/1e6:.1f} M updates/s)')
