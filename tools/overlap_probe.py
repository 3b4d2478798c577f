#!/usr/bin/env python3
"""Probe: a P1 full-output step as TWO INDEPENDENT kernels on two streams — p1_point_kernel (s, ds, ind_p, branch counters)
beside the one-kernel K,F step (p1_fused_kernel, which evaluates the return map of its staged elements itself and reads
none of the point kernel's outputs) — against the default two dependent kernels.  Same mesh and field as bench.py."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

fep = importlib.import_module('fem-elastoplasticity_amd')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 708
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda', 0)
mesh = fep.rect_mesh(N, N, 'P1', 10.0, 10.0)
sh = bench.Shard(fep, torch, mesh, 0, 1, dev, 1.0, False)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
ctx = sh.ctx


def serial():
    sh.step(main.cuda_stream, 0, True)


ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()


def overlapped():
    ev_fork.record(main)
    side.wait_event(ev_fork)
    ctx.step_dev(side.cuda_stream, sh.U.data_ptr(), ep=sh.Ep.data_ptr(), accept=False, s=sh.S.data_ptr(), ds=sh.DS.data_ptr(),
                 ind_p=sh.indp.data_ptr(), counts=sh.counts.data_ptr())
    ctx.step_dev(main.cuda_stream, sh.U.data_ptr(), ep=sh.Ep.data_ptr(), accept=False, k_data=sh.Kd.data_ptr(), f_out=sh.Fb[0].data_ptr())
    ev_join.record(side)
    main.wait_event(ev_join)


def timeit(f):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


serial(); torch.cuda.synchronize()
ref = (sh.Kd.clone(), sh.Fb[0].clone(), sh.S.clone(), sh.DS.clone(), sh.indp.clone(), sh.counts.clone())
sh.Kd.zero_(); sh.Fb[0].zero_(); sh.S.zero_(); sh.DS.zero_(); sh.indp.zero_()
overlapped(); torch.cuda.synchronize()
same = [bool(torch.equal(a, b)) for a, b in zip(ref, (sh.Kd, sh.Fb[0], sh.S, sh.DS, sh.indp, sh.counts))]
print('outputs bit-identical (K, F, s, ds, ind_p, counts):', same)
for rep in range(3):
    print(f'serial two kernels {timeit(serial):.4f} ms   overlapped {timeit(overlapped):.4f} ms', flush=True)
