#!/usr/bin/env python3
"""Per-element-type timing of the hot path (in-situ HIP-event timings per kernel).
    python tools/elem_bench.py P2 354 [steps]
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
t = sys.argv[1]
N = int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mix = sys.argv[4] if len(sys.argv) > 4 else 'bands'
kf_only = len(sys.argv) > 5 and sys.argv[5] == 'kf'      # what a Newton iterate asks for: K and F, no s / ds / ind_p
t0 = time.time()
if t == 'P4':                       # P4 meshes come from the midpoint generator (TSX:1354-1505) on a P1 mesh
    m1 = fep.square_mesh(N, 'P1', 10)
    mp = fep.create_midpoints_P4(m1['coordinates'], m1['elements'])
    mesh = {'elements': np.asarray(mp['elem_ext'], dtype=np.int64), 'coordinates': mp['coord_ext']}
else:
    mesh = fep.square_mesh(N, t, 10)
ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
t_setup = time.time() - t0
ctx.set_materials(*bench.dp_materials())
dev = torch.device('cuda', 0)
Uh = bench.displacement(mesh['coordinates'])
if mix == 'random':          # i.i.d. branch per point: worst-case divergence inside a wave (config 5)
    Uh = Uh + np.random.default_rng(5).normal(0, 1.5e-4 * 10 / N, size=Uh.shape)
U = torch.from_numpy(np.ascontiguousarray(Uh.reshape(-1, order='F'))).to(dev)
n = ctx.n_int
f64 = dict(dtype=torch.float64, device=dev)
Ep = torch.zeros((4, n), **f64); S = torch.empty((4, n), **f64); DS = torch.empty((9, n), **f64)
ind = torch.empty(n, dtype=torch.uint8, device=dev); Kd = torch.empty(ctx.nnz, **f64); F = torch.empty(ctx.n_dof, **f64)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
cptr = 0 if os.environ.get('FEP_BENCH_NOCOUNTS') else cnt.data_ptr()


def step():
    if kf_only:
        ctx.step_dev(st, U.data_ptr(), ep=Ep.data_ptr(), k_data=Kd.data_ptr(), f_out=F.data_ptr(), counts=cptr)
    else:
        ctx.step_dev(st, U.data_ptr(), ep=Ep.data_ptr(), s=S.data_ptr(), ds=DS.data_ptr(), ind_p=ind.data_ptr(),
                     k_data=Kd.data_ptr(), f_out=F.data_ptr(), counts=cptr)


for _ in range(3):
    step()
torch.cuda.synchronize()
dt = 1e9
for _ in range(3):                  # best of three batches (box-to-box and run-to-run spread is several per cent)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = min(dt, (time.perf_counter() - t0) / steps)
ctx.profile_begin()
for _ in range(steps):
    step()
kms, _ = ctx.profile_end(st)
n_p, n_q = fep.ELEMENT_SHAPE[fep.LagrangeElementType[t]]
alg = (201 + 16 * n_p + 8 * (2 * n_p) ** 2 / n_q) * n
c = cnt.cpu().tolist()
print(f'{t}{" K/F-only" if kf_only else ""} N={N} n_e={ctx.n_e} n_int={n} nnz={ctx.nnz} setup {t_setup:.1f}s  smooth/apex {c[0]}/{c[1]}  '
      f'step {dt*1e3:.3f} ms -> {n/dt/1e9:.2f} G upd/s | kernels ms {({k: round(v, 4) for k, v in kms.items()})} | '
      f'alg {alg/1e6:.0f} MB -> element kernel {alg/(kms["element"]*1e-3)/1e12:.2f} TB/s, '
      f'whole step {alg/dt/1e12:.2f} TB/s')
if os.environ.get('FEP_BENCH_HASH'):        # bitwise comparison of K and F between libraries (same inputs): one digest per run
    import hashlib
    print('  sha1 K', hashlib.sha1(Kd.cpu().numpy().tobytes()).hexdigest()[:16], 'F', hashlib.sha1(F.cpu().numpy().tobytes()).hexdigest()[:16])
ctx.close()                         # (the ablation build reports its phase clocks when the context is destroyed)
