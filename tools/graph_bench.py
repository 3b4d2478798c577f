#!/usr/bin/env python3
"""Plain launches against a captured HIP graph for one step of the hot path (bench workload)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
mesh = fep.square_mesh(708, 'P1', 10)
ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
ctx.set_materials(*bench.dp_materials())
dev = torch.device('cuda', 0)
f64 = dict(dtype=torch.float64, device=dev)
n = ctx.n_int
U = torch.from_numpy(np.ascontiguousarray(bench.displacement(mesh['coordinates']).reshape(-1, order='F'))).to(dev)
Ep = torch.zeros((4, n), **f64); S = torch.empty((4, n), **f64); DS = torch.empty((9, n), **f64)
ind = torch.empty(n, dtype=torch.uint8, device=dev); Kd = torch.empty(ctx.nnz, **f64); F = torch.empty(ctx.n_dof, **f64)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)


def step():
    ctx.step_dev(torch.cuda.current_stream().cuda_stream, U.data_ptr(), ep=Ep.data_ptr(), s=S.data_ptr(), ds=DS.data_ptr(),
                 ind_p=ind.data_ptr(), k_data=Kd.data_ptr(), f_out=F.data_ptr(), counts=cnt.data_ptr())


def timed(fn, k=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
g10 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g10):
    for _ in range(10):
        step()
print(f'plain launches {timed(step):.4f} ms/step, graph of 1 step {timed(g.replay):.4f} ms/step, '
      f'graph of 10 steps {timed(g10.replay, 30) / 10:.4f} ms/step')
