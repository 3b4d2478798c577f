#!/usr/bin/env python3
"""Probe: what would the element types' step cost as TWO kernels, the way the P1 route does it — the mesh-free return map over the
points (fep_return_map_dev on a given strain: streams at the P1 point kernel's rate) + the assembly from ds / s (fep_assemble_dev:
element_kernel<FROM_U = false> + fixup_kernel) — against the one-kernel step?   python tools/split_probe.py P2 708 [steps]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
_lib = importlib.import_module('fem-elastoplasticity_amd._lib')
t, N = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if t == 'P4':
    m1 = fep.square_mesh(N, 'P1', 10)
    mp = fep.create_midpoints_P4(m1['coordinates'], m1['elements'])
    mesh = {'elements': np.asarray(mp['elem_ext'], dtype=np.int64), 'coordinates': mp['coord_ext']}
else:
    mesh = fep.square_mesh(N, t, 10)
ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
ctx.set_materials(*bench.dp_materials())
dev = torch.device('cuda', 0)
U = torch.from_numpy(np.ascontiguousarray(bench.displacement(mesh['coordinates']).reshape(-1, order='F'))).to(dev)
n = ctx.n_int
f64 = dict(dtype=torch.float64, device=dev)
Ep = torch.zeros((4, n), **f64); S = torch.empty((4, n), **f64); DS = torch.empty((9, n), **f64); E = torch.empty((3, n), **f64)
ind = torch.empty(n, dtype=torch.uint8, device=dev); Kd = torch.empty(ctx.nnz, **f64); F = torch.empty(ctx.n_dof, **f64)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
mats = [torch.full((n,), v, **f64) for v in bench.dp_materials()]
st = torch.cuda.current_stream().cuda_stream
l = _lib.lib()


def full():
    ctx.step_dev(st, U.data_ptr(), ep=Ep.data_ptr(), s=S.data_ptr(), ds=DS.data_ptr(), ind_p=ind.data_ptr(), k_data=Kd.data_ptr(),
                 f_out=F.data_ptr(), counts=cnt.data_ptr())


def rmap():
    _lib.check(l.fep_return_map_dev(0, st, n, E.data_ptr(), 1, n, None, Ep.data_ptr(), mats[0].data_ptr(), mats[1].data_ptr(),
                                    mats[2].data_ptr(), mats[3].data_ptr(), 0, S.data_ptr(), DS.data_ptr(), ind.data_ptr(), cnt.data_ptr()),
               'fep_return_map_dev')


def asm():
    ctx.assemble_dev(st, ds=DS.data_ptr(), s=S.data_ptr(), k_data=Kd.data_ptr(), f_out=F.data_ptr())


def timeit(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            f()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3


ctx.step_dev(st, U.data_ptr(), ep=Ep.data_ptr(), e_out=E.data_ptr(), s=S.data_ptr())
a, b, c = timeit(full), timeit(rmap), timeit(asm)
print(f'{t} N={N} n_e={ctx.n_e}: one-kernel step {a:.3f} ms | mesh-free return map on the strain {b:.3f} ms + assembly from ds, s {c:.3f} ms = {b + c:.3f} ms '
      f'(the split step would also have to form the strain: + the gather of U)')
ctx.close()
