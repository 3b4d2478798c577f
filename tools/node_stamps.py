#!/usr/bin/env python3
"""Diagnostic: where does a workgroup of the P1 assembly kernel spend its cycles?  Runs the stamped
build (fep_debug_p1_node_stamps) on the bench workload and prints per-phase cycle statistics."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
import torch  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 708
mesh = fep.square_mesh(N, 'P1', 10)
ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
ctx.set_materials(*bench.dp_materials())
dev = torch.device('cuda', 0)
U = torch.from_numpy(np.ascontiguousarray(bench.displacement(mesh['coordinates']).reshape(-1, order='F'))).to(dev)
n = ctx.n_int
S = torch.empty((4, n), dtype=torch.float64, device=dev)
DS = torch.empty((9, n), dtype=torch.float64, device=dev)
Kd = torch.empty(ctx.nnz, dtype=torch.float64, device=dev)
F = torch.empty(ctx.n_dof, dtype=torch.float64, device=dev)
ctx.step_dev(0, U.data_ptr(), s=S.data_ptr(), ds=DS.data_ptr(), k_data=Kd.data_ptr(), f_out=F.data_ptr())
torch.cuda.synchronize()
l = fep.lib()
n_wg = ctx.n_blk // 128 + 8          # upper bound; the library returns the real tile count
st = np.zeros((n_wg, 8), dtype=np.int64)
nw = C.c_int()
fn = l.fep_debug_p1_node_stamps
fn.restype = C.c_int
rc = fn(ctx.handle, C.c_void_p(DS.data_ptr()), C.c_void_p(S.data_ptr()), C.c_void_p(Kd.data_ptr()), C.c_void_p(F.data_ptr()),
        st.ctypes.data_as(C.c_void_p), C.c_int64(st.size), C.byref(nw))
assert rc == 0, rc
st = st[:nw.value]
n_wg = nw.value
d = np.diff(st[:, :6], axis=1)
names = ['prologue loads (segptr/meta/codes)', 'eptr + list entry', 'element data -> LDS', 'barrier wait', 'gather+FMA+store']
print(f'n_wg={n_wg}  staged elements per wg: mean {st[:,7].mean():.1f} max {st[:,7].max()}')
for k, nm in enumerate(names):
    print(f'{nm:38s} mean {d[:,k].mean():9.0f}  p50 {np.median(d[:,k]):9.0f}  p90 {np.percentile(d[:,k],90):9.0f} cycles')
tot = st[:, 5] - st[:, 0]
print(f'{"workgroup lifetime":38s} mean {tot.mean():9.0f}  p50 {np.median(tot):9.0f}  p90 {np.percentile(tot,90):9.0f} cycles')
span = st[:, 5].max() - st[:, 0].min()
print(f'kernel span {span} ticks (s_memtime, 100 MHz?)  sum(lifetime)/span = {tot.sum()/span:.1f} wgs in flight on average')
