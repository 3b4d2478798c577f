#!/usr/bin/env python3
"""
Benchmark of the hot path: element*quadrature-point updates per second for
strain -> Drucker-Prager return map -> tangent-stiffness (CSR values) -> internal force.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[3], the configuration the metric's target is quoted on): the
strip-footing square refined to 708 x 708 cells = 1 002 528 P1 elements (= integration points) per
GPU, Drucker-Prager materials of the reference demo (DP:910-933), a synthetic displacement field
that puts points on all three branches (elastic / smooth / apex).  One step = one pass of the hot
path with every input resident in HBM.  With N GPUs the mesh is a 708 x 708N rectangle sharded by
contiguous element ranges (weak scaling); the only exchange is the RCCL all-reduce of the
interface forces.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (fused element
kernel, HIP-event timed on its stream, algorithmic bytes 537 B/update for P1: SURVEY 8d) and
`cpu_baseline` (the NumPy oracle timed on this host, N=1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CELLS = 708                       # 2*708^2 = 1 002 528 P1 elements per GPU
HBM_PEAK_GBS = 8000.0               # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s)
ALG_BYTES = {'P1': 201 + 16 * 3 + 8 * 36 / 1, 'P2': 201 + 16 * 6 + 8 * 144 / 7,
             'Q1': 201 + 16 * 4 + 8 * 64 / 4, 'Q2': 201 + 16 * 8 + 8 * 256 / 9}


def dp_materials():
    young, nu, c0, phi = 1e7, 0.48, 450, np.pi / 9                       # DP:910-933
    return (young / (2 * (1 + nu)), young / (3 * (1 - 2 * nu)),
            3 * np.tan(phi) / np.sqrt(9 + 12 * np.tan(phi) ** 2), 3 * c0 / np.sqrt(9 + 12 * np.tan(phi) ** 2))


def displacement(coord, seed=1):
    """Synthetic state: shear/compression bands + noise, periodic in y with the strip height."""
    x, y = coord[0], np.mod(coord[1], 10.0)
    U = np.array([2.0e-4 * y * (x / 10) + 1.0e-4 * x * (y > 5), -1.2e-4 * y * (x < 5) + 1.6e-4 * y * (x >= 5)])
    U += np.random.default_rng(seed).normal(0, 2e-8, size=U.shape)
    return U


def cpu_baseline(fep, elem_type='P1', n_cells=354, repeats=3):
    """The oracle (NumPy/SciPy restatement of the reference path) on a bounded sample of the same
    workload: a quarter-size square with the same field and materials, 1 host thread."""
    from oracle import fep_oracle as orc
    mesh = fep.square_mesh(n_cells, elem_type, 10)
    elem, coord = mesh['elements'], mesh['coordinates']
    d1, d2, wf = fep.element_tables(elem_type)
    n_int = elem.shape[1] * wf.size
    G, Kb, eta, c = dp_materials()
    one = np.ones(n_int)
    K, B, w, iD, jD, D = orc.elastic_setup(elem, coord, G * one, Kb * one, d1, d2, wf)
    ctx = dict(K_elast=K, B=B, D_elast=D, weight=w, iD=iD, jD=jD, shear=G * one, bulk=Kb * one, eta=eta * one, c=c * one)
    U = displacement(coord)
    Ep = np.zeros((4, n_int))
    best = float('inf')
    for _ in range(repeats):
        t0 = time.perf_counter()
        orc.hot_path(U, Ep, ctx)
        best = min(best, time.perf_counter() - t0)
    return {'value': n_int / best, 'unit': 'updates/s', 'cores': 1, 'kind': 'port',
            'sample': f'oracle.hot_path (NumPy/SciPy, single thread) on a {n_cells}x{n_cells}-cell {elem_type} square '
                      f'({n_int} points, same field/materials), best of {repeats}, {best:.3f} s/pass; '
                      f'host has {os.cpu_count()} cores'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--cells', type=int, default=N_CELLS, help='cells per side of the per-GPU square')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run')
    if os.environ.get('FEP_BENCH_SINGLE_DEVICE'):      # rehearsal of the N>1 path on a one-GPU box: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    fep = importlib.import_module('fem-elastoplasticity_amd')
    fep.lib()                                      # fails loudly if the HIP extension is missing
    N = args.cells
    mesh = fep.rect_mesh(N, N * world, 'P1', 10, 10 * world)
    sh = fep.ShardedContext(mesh['elements'], mesh['coordinates'], rank, world, device=local_rank)
    ctx = sh.ctx
    G, Kb, eta, c = dp_materials()
    ctx.set_materials(G, Kb, eta, c)
    n_int = ctx.n_int
    U_h = displacement(mesh['coordinates'][:, sh.nodes])
    f64 = dict(dtype=torch.float64, device=dev)
    U = torch.from_numpy(np.ascontiguousarray(U_h.reshape(-1, order='F'))).to(dev)
    Ep = torch.zeros((4, n_int), **f64)
    S = torch.empty((4, n_int), **f64)
    DS = torch.empty((9, n_int), **f64)
    indp = torch.empty(n_int, dtype=torch.uint8, device=dev)
    Kd = torch.empty(ctx.nnz, **f64)
    Fb = [torch.empty(ctx.n_dof, **f64) for _ in range(2 if world > 1 else 1)]
    F = Fb[0]
    counts = torch.zeros(2, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    # N > 1: the interface all-reduce of pass i runs on a second stream under pass i+1's kernels (the force
    # vector is double-buffered; a pass only waits for the exchange that last used its buffer)
    main = torch.cuda.current_stream()
    comm = torch.cuda.Stream() if world > 1 else None
    ev_done = [torch.cuda.Event() for _ in range(2)]
    ev_ready = [torch.cuda.Event() for _ in range(2)]
    state = {'i': 0}

    def step():
        i = state['i'] & 1
        state['i'] += 1
        Fi = Fb[i] if world > 1 else F
        if world > 1 and state['i'] > 2:
            main.wait_event(ev_done[i])
        ctx.step_dev(stream, U.data_ptr(), ep=Ep.data_ptr(), accept=False, s=S.data_ptr(), ds=DS.data_ptr(),
                     ind_p=indp.data_ptr(), k_data=Kd.data_ptr(), f_out=Fi.data_ptr(), counts=counts.data_ptr())
        if world > 1:
            ev_ready[i].record(main)
            comm.wait_event(ev_ready[i])
            with torch.cuda.stream(comm):
                sh.exchange_force_(Fi)
                ev_done[i].record(comm)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], **f64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    n_smooth, n_apex = [int(v) for v in counts.cpu()]

    # per-kernel durations IN SITU: the same step sequence again with HIP events around every kernel
    # on the launch stream (not part of the timed region above)
    ctx.profile_begin()
    for _ in range(args.steps):
        step()
    kms, n_prof = ctx.profile_end(stream)
    barrier()
    route = os.environ.get('FEP_P1_PATH', '') or 'node'
    if route == 'coo':
        # fused element kernel: strain + return map + K_e/f_e blocks (exactly the work SURVEY 8d prices)
        k_ms = kms['element']
        k_name = 'element_kernel<3,1,true> (strain + return map + K_e/f_e)'
        others = {'csr_reduce_kernel': kms['csr'], 'force_reduce_kernel': kms['force']}
        per_kernel = None
    else:
        # node route: the priced work (return map + tangent assembly) is the PAIR of kernels; the
        # figure divides SURVEY 8d's bytes by the SUM of both durations (which also includes the CSR
        # numeric phase and the force gather that 8d prices separately)
        k_ms = kms['element'] + kms['csr']
        k_name = 'p1_point_kernel + p1_node_kernel (strain + return map; tangent CSR values + force)'
        others = {'p1_point_kernel': kms['element'], 'p1_node_kernel': kms['csr']}
        # each kernel against its OWN minimal HBM bytes: point = elem ids 12 + ep 32
        # + coordinates/displacements 32 (16 B per node, ~2 elements per node) + s 32 + ds 72 + ind_p 1 = 181 B (the
        # materials are constant over the mesh here and are not read: 32 B less than the general case);
        # assembly = ds 48 (6 of 9 rows) + s 24 + geometry record 48 + descriptors/codes 32 + CSR values 8*nnz/n + force 16*n_n/n
        b_point = 181.0 * n_int
        b_node = (48 + 24 + 48 + 32) * n_int + 8.0 * ctx.nnz + 8.0 * ctx.n_dof
        per_kernel = {'p1_point_kernel': {'bytes': b_point, 'GBps': b_point / (kms['element'] * 1e-3) / 1e9,
                                          'frac': b_point / (kms['element'] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                      'p1_node_kernel': {'bytes': b_node, 'GBps': b_node / (kms['csr'] * 1e-3) / 1e9,
                                         'frac': b_node / (kms['csr'] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    alg = ALG_BYTES['P1'] * n_int
    achieved = alg / (k_ms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get('elements_per_gpu', 1002528) == n_int and route == 'node':
                    traffic = tj.get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        line = {
            'metric': 'element*quadpt updates/sec (return-map + K_tan assemble)',
            'value': world * n_int * args.steps / dt, 'unit': 'updates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'Plasticity2D_DP strip-footing square, {N}x{N} cells = {n_int} P1 elements per GPU '
                                   f'(BASELINE configs[3]), Drucker-Prager, strain->return map->K_tan CSR values->F per step',
                       'elements_per_gpu': n_int, 'nnz_per_gpu': ctx.nnz, 'smooth_points': n_smooth, 'apex_points': n_apex,
                       'route': route,
                       'parallelism': f'element-shard x{world}, interface-force all-reduce' if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': k_name, 'kernel_ms': k_ms,
                         'algorithmic_bytes_per_launch': alg, 'timing': f'HIP events in situ, mean of {n_prof} launches',
                         'kernels_ms': others, 'per_kernel': per_kernel},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(fep)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
