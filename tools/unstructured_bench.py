#!/usr/bin/env python3
"""Hot path on an UNSTRUCTURED P1 mesh (Delaunay triangulation of jittered points): which route the context picks,
tile statistics and step time, for different node / element numberings.
    python tools/unstructured_bench.py [n_points_per_side=708] [order=morton|random|rows] [renumber]"""
import importlib
import os
import sys
import time

import numpy as np
from scipy.spatial import Delaunay

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 708
order = sys.argv[2] if len(sys.argv) > 2 else 'morton'
rng = np.random.default_rng(11)
g = (np.stack(np.meshgrid(np.arange(M + 1), np.arange(M + 1), indexing='xy')).reshape(2, -1).astype(float))
inner = (g[0] > 0) & (g[0] < M) & (g[1] > 0) & (g[1] < M)
g[:, inner] += rng.uniform(-0.35, 0.35, size=(2, int(inner.sum())))
pts = g * (10.0 / M)


def morton(ix, iy):
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        for s, m in ((16, 0x0000FFFF0000FFFF), (8, 0x00FF00FF00FF00FF), (4, 0x0F0F0F0F0F0F0F0F), (2, 0x3333333333333333),
                     (1, 0x5555555555555555)):
            v = (v | (v << np.uint64(s))) & np.uint64(m)
        return v
    return spread(ix) | (spread(iy) << np.uint64(1))


q = np.floor(pts / 10.0 * 65535).astype(np.int64)
if order == 'morton':
    perm = np.argsort(morton(q[0], q[1]), kind='stable')
elif order == 'random':
    perm = rng.permutation(pts.shape[1])
else:                                                   # rows: the generator's row-major order
    perm = np.arange(pts.shape[1])
pts = pts[:, perm]
t0 = time.time()
tri = Delaunay(pts.T).simplices.T.astype(np.int64)      # (3, n_e)
cen = pts[:, tri].mean(axis=1)
if order == 'random':
    eperm = rng.permutation(tri.shape[1])
else:                                                   # elements follow their lowest node (what a mesher that numbers nodes first does)
    eperm = np.argsort(tri.min(axis=0), kind='stable')
tri = tri[:, eperm]
if len(sys.argv) > 3 and sys.argv[3] == 'renumber':          # what a caller with a badly numbered mesh should do first
    tri, pts, _, _ = fep.renumber_for_locality(tri, pts)
t_mesh = time.time() - t0
t0 = time.time()
ctx = fep.MeshContext(tri, pts)
t_ctx = time.time() - t0
ctx.set_materials(*bench.dp_materials())
dev = torch.device('cuda', 0)
f64 = dict(dtype=torch.float64, device=dev)
n = ctx.n_int
U = torch.from_numpy(np.ascontiguousarray(bench.displacement(pts).reshape(-1, order='F'))).to(dev)
Ep = torch.zeros((4, n), **f64); S = torch.empty((4, n), **f64); DS = torch.empty((9, n), **f64)
ind = torch.empty(n, dtype=torch.uint8, device=dev); Kd = torch.empty(ctx.nnz, **f64); F = torch.empty(ctx.n_dof, **f64)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def step():
    ctx.step_dev(st, U.data_ptr(), ep=Ep.data_ptr(), s=S.data_ptr(), ds=DS.data_ptr(), ind_p=ind.data_ptr(),
                 k_data=Kd.data_ptr(), f_out=F.data_ptr(), counts=cnt.data_ptr())


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
ctx.profile_begin()
for _ in range(20):
    step()
kms, _ = ctx.profile_end(st)
c = cnt.cpu().tolist()
print(f'order={order}: {ctx.n_e} P1 elements, {ctx.n_n} nodes, nnz {ctx.nnz}; Delaunay {t_mesh:.1f} s, context {t_ctx:.2f} s; '
      f'smooth/apex {c[0]}/{c[1]}; step {dt*1e3:.4f} ms = {n/dt/1e9:.2f} G updates/s; kernels ms '
      f'{({k: round(v, 4) for k, v in kms.items()})}')
