#!/usr/bin/env python3
"""How long does the HOST need per step (Python + ctypes + launches), with and without the interface exchange
path (all-reduce itself skipped)?  If this exceeds the GPU time per step the multi-GPU bench is host-bound."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
fep = importlib.import_module('fem-elastoplasticity_amd')
N = 708
mesh = fep.rect_mesh(N, 2 * N, 'P1', 10, 20)
sh = fep.ShardedContext(mesh['elements'], mesh['coordinates'], 0, 2, device=0)
sh.world = 1                                    # keep pack/unpack, skip the collective
ctx = sh.ctx
ctx.set_materials(*bench.dp_materials())
dev = torch.device('cuda', 0)
U = torch.from_numpy(np.ascontiguousarray(bench.displacement(mesh['coordinates'][:, sh.nodes]).reshape(-1, order='F'))).to(dev)
n = ctx.n_int
f64 = dict(dtype=torch.float64, device=dev)
Ep = torch.zeros((4, n), **f64); S = torch.empty((4, n), **f64); DS = torch.empty((9, n), **f64)
ind = torch.empty(n, dtype=torch.uint8, device=dev); Kd = torch.empty(ctx.nnz, **f64); F = torch.empty(ctx.n_dof, **f64)
cnt = torch.zeros(2, dtype=torch.int64, device=dev)
main = torch.cuda.current_stream(); comm = torch.cuda.Stream(); st = main.cuda_stream
ev = [torch.cuda.Event() for _ in range(4)]
def step(ex):
    ctx.step_dev(st, U.data_ptr(), ep=Ep.data_ptr(), s=S.data_ptr(), ds=DS.data_ptr(), ind_p=ind.data_ptr(),
                 k_data=Kd.data_ptr(), f_out=F.data_ptr(), counts=cnt.data_ptr())
    if ex:
        ev[0].record(main); comm.wait_event(ev[0])
        with torch.cuda.stream(comm):
            sh.exchange_force_(F); ev[1].record(comm)
for ex in (False, True):
    for _ in range(10): step(ex)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K): step(ex)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'exchange path {ex}: host issue time {t_host/K*1e6:.1f} us/step, wall {t_all/K*1e6:.1f} us/step (n_iface={sh.n_iface})')
