#!/usr/bin/env python3
"""
Benchmark of the hot path: element*quadrature-point updates per second for
strain -> Drucker-Prager return map -> tangent-stiffness (CSR values) -> internal force.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--elem P1|P2|Q1|Q2] [--cells C] [--state S]

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` as a
CHILD process, before anything in this process touches the GPU), waits and returns the child's exit code; launched
under torch.distributed.run (WORLD_SIZE set) it is one rank.

Workload (BASELINE.json configs[3], the configuration the metric's target is quoted on): the strip-footing
square refined to 708 x 708 cells = 1 002 528 P1 elements (= integration points), Drucker-Prager materials of
the reference demo (DP:910-933), a synthetic displacement field that puts points on all three branches
(elastic / smooth / apex).  One step = one pass of the hot path with every input resident in HBM.
  --scaling weak   (default) every GPU holds its own 708 x 708 square of a 708 x 708N rectangle;
  --scaling strong ONE 708 x 708 square split N ways (configs[3] as written: "1 vs 2 vs 4 vs 8 GPUs").
Either way the mesh is sharded by contiguous element ranges, the hot path has no data-path collective and the
only exchange is the RCCL all-reduce of the interface forces (one node row per cut).  At N > 1 the weak line
also carries a `strong` object: the same K steps on the one-square mesh split N ways.

`--elem P2` (default 1414 cells = 3 998 792 elements = 27 991 544 points; `--state random` puts the three branches i.i.d.
on the points) is BASELINE configs[4]; its documented 8-GPU command is
    python bench.py --gpus 8 --elem P2 --cells 1414 --state random --scaling strong
(one square split over the ranks by contiguous element ranges; the element types other than P1 run the element route).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HIP-event timed on the launch
stream, algorithmic bytes 537 B/update for P1: SURVEY 8d) and `cpu_baseline` (the NumPy oracle timed on this
host, N=1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_CELLS = 708                       # 2*708^2 = 1 002 528 P1 elements
DEFAULT_CELLS = {'P1': 708, 'P2': 1414, 'Q1': 708, 'Q2': 708}          # P2: BASELINE configs[4] (3 998 792 elements)
NQ = {'P1': 1, 'P2': 7, 'Q1': 4, 'Q2': 9}
HBM_PEAK_GBS = 8000.0               # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s)
ALG_BYTES = {'P1': 201 + 16 * 3 + 8 * 36 / 1, 'P2': 201 + 16 * 6 + 8 * 144 / 7,
             'Q1': 201 + 16 * 4 + 8 * 64 / 4, 'Q2': 201 + 16 * 8 + 8 * 256 / 9,
             'P4': 201 + 16 * 15 + 8 * 900 / 12}


def dp_materials():
    young, nu, c0, phi = 1e7, 0.48, 450, np.pi / 9                       # DP:910-933
    return (young / (2 * (1 + nu)), young / (3 * (1 - 2 * nu)),
            3 * np.tan(phi) / np.sqrt(9 + 12 * np.tan(phi) ** 2), 3 * c0 / np.sqrt(9 + 12 * np.tan(phi) ** 2))


def displacement(coord, seed=1, scale=1.0, state='bands'):
    """Synthetic state on the GLOBAL node set, periodic in y with the strip height, + noise.
    'bands'  (default) shear / compression / tension bands: ~18 % smooth and ~57 % apex points;
    'random' the same with node noise of the size of the field's increments: the three branches i.i.d. over the points;
    'newton' a branch mix like the Newton iterates of configs[3] (tools/newton_bench.py: ~30 % smooth, < 0.1 % apex):
             slight uniform compression with a strong shear in the strip x < 3 (29 % smooth, no apex point)."""
    x, y = coord[0], np.mod(coord[1], 10.0)
    if state == 'newton':
        U = scale * np.array([np.where(x < 3.0, 4.0e-4, 0.8e-4) * y, -2.0e-5 * y])
    else:
        U = scale * np.array([2.0e-4 * y * (x / 10) + 1.0e-4 * x * (y > 5), -1.2e-4 * y * (x < 5) + 1.6e-4 * y * (x >= 5)])
    if state == 'random':           # i.i.d. branch per point: worst-case divergence inside a wave (configs[4]); the noise
        n_side = max(1.0, np.sqrt(coord.shape[1]) - 1)       # amplitude follows the node spacing so that strains stay O(1e-4)
        U = U + np.random.default_rng(5).normal(0, 1.5e-4 * 10 / n_side, size=U.shape)
    U += np.random.default_rng(seed).normal(0, 2e-8, size=U.shape)
    return U


def cpu_baseline(fep, elem_type='P1', n_cells=N_CELLS, repeats=3, scale=1.0, state='bands'):
    """The oracle (NumPy/SciPy restatement of the reference path) on the same workload (same mesh, field and
    materials), 1 host thread."""
    from oracle import fep_oracle as orc
    mesh = fep.square_mesh(n_cells, elem_type, 10)
    elem, coord = mesh['elements'], mesh['coordinates']
    d1, d2, wf = fep.element_tables(elem_type)
    n_int = elem.shape[1] * wf.size
    G, Kb, eta, c = dp_materials()
    one = np.ones(n_int)
    K, B, w, iD, jD, D = orc.elastic_setup(elem, coord, G * one, Kb * one, d1, d2, wf)
    ctx = dict(K_elast=K, B=B, D_elast=D, weight=w, iD=iD, jD=jD, shear=G * one, bulk=Kb * one, eta=eta * one, c=c * one)
    U = displacement(coord, scale=scale, state=state)
    Ep = np.zeros((4, n_int))
    best = float('inf')
    out = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        out = orc.hot_path(U, Ep, ctx)
        best = min(best, time.perf_counter() - t0)
    return {'value': n_int / best, 'unit': 'updates/s', 'cores': 1, 'kind': 'port',
            'sample': f'oracle.hot_path (NumPy/SciPy, single thread) on the {n_cells}x{n_cells}-cell {elem_type} square '
                      f'({n_int} points, same field/materials), best of {repeats}, {best:.3f} s/pass; '
                      f'host has {os.cpu_count()} cores'}, (out, K)


def parity_vs_oracle(oracle_out, shard, torch):
    """The checker's result of the cpu_baseline leg against what the GPU step left in the shard's buffers (same mesh, same
    field): largest error of s, ds (relative to the array's largest entry), every row of K (relative to the row's largest
    entry OR of the same row of K_elast, whichever is larger: the oracle forms K_elast + B^T (D_p - D_elast) B as the
    reference does (DP:1050), so a row whose points are all at the apex is a difference of two numbers of K_elast's size
    there — 1e-9 of round-off on an exact zero), F; ind_p must agree exactly.  Outside the timed region."""
    import scipy.sparse as ssp
    (E, cp, K_t, F), K_el = oracle_out
    S = shard.S.cpu().numpy(); DS = shard.DS.cpu().numpy(); ind = shard.indp.cpu().numpy().astype(bool)
    ip, ix = shard.ctx.pattern()
    K = ssp.csr_matrix((shard.Kd.cpu().numpy(), ix, ip), shape=(shard.ctx.n_dof, shard.ctx.n_dof))

    def rel(a, b):
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
    D = abs(K - K_t).tocsr()
    d = np.asarray(D.max(axis=1).todense()).ravel()
    sc = np.maximum(np.asarray(abs(K_t).max(axis=1).todense()).ravel(), np.asarray(abs(K_el).max(axis=1).todense()).ravel())
    k_rows = float(np.where(sc > 0, d / np.where(sc > 0, sc, 1.0), d).max())
    F_g = shard.Fb[0].cpu().numpy()
    errs = {'s': rel(S, cp['s']), 'ds': rel(DS, cp['ds']), 'K': float(d.max() / abs(K_t).max()), 'K_rows': k_rows,
            'F': rel(F_g, np.asarray(F).ravel()),
            'ind_p_mismatches': int(np.count_nonzero(ind != np.asarray(cp['ind_p']).ravel()))}
    return errs


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--elem', choices=('P1', 'P2', 'Q1', 'Q2'), default='P1',
                    help='element type: P1 = BASELINE configs[3] (default), P2 = configs[4]')
    ap.add_argument('--cells', type=int, default=None,
                    help='cells per side of the square (per GPU when weak); default 708 (P1, Q1, Q2), 1414 (P2)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--field-scale', type=float, default=1.0, help='multiplies the displacement field')
    ap.add_argument('--state', choices=('bands', 'newton', 'random'), default='bands',
                    help="branch mix of the synthetic state: 'bands' ~18 %% smooth / 57 %% apex (default), 'newton' ~29 %% "
                         "smooth / no apex, like the Newton iterates of configs[3]")
    ap.add_argument('--preheat-ms', type=float, default=250.0,
                    help='untimed passes of the same step for this long before the W warm-up steps of every timed run (the '
                         'clocks ramp with load: a 2 ms timed region straight after set-up measures 4-5 %% slower); 0 = off')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-cells', type=int, default=None,
                    help='cells per side of the CPU-baseline square (default: --cells for P1, min(--cells, 354) otherwise: a bounded sample)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)')
    ap.add_argument('--exchange', choices=('allreduce', 'p2p'), default=os.environ.get('FEP_EXCHANGE', 'allreduce'),
                    help='form of the interface exchange the HEADLINE uses at N > 1 (both are timed in the same run: `exchange`)')
    ap.add_argument('--min-elements-per-rank', type=int, default=100000,
                    help='the north_star gate: only as many ranks take elements as leaves each at least this many (0 = off)')
    return ap.parse_args()


def launch_ranks(n):
    """Fresh child process running torch.distributed.run with n ranks of this script; nothing in THIS process has
    touched the GPU (torch is not even imported yet).  Returns the child's exit code."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class Shard:
    """One rank's part of a mesh: device context + device-resident state of the benchmark step."""

    def __init__(self, fep, torch, mesh, rank, world, dev, scale, two_buffers, state='bands', exchange='allreduce', min_elements=0):
        self.sh = fep.ShardedContext(mesh['elements'], mesh['coordinates'], rank, world, device=dev.index,   # element type from n_p
                                     min_elements_per_rank=min_elements, exchange=exchange)
        ctx = self.ctx = self.sh.ctx
        self.idle = ctx is None                         # a rank the gate left without elements: no kernels, zeros into the all-reduce
        if self.idle:
            self.n_int = 0
            self.Fb = [torch.empty(0, dtype=torch.float64, device=dev) for _ in range(2 if two_buffers else 1)]
            self.counts = torch.zeros(2, dtype=torch.int64, device=dev)
            return
        ctx.set_materials(*dp_materials())
        n_int = self.n_int = ctx.n_int
        # the field is a function of the GLOBAL node set (identical on every rank that shares a node)
        U_h = displacement(mesh['coordinates'], scale=scale, state=state)[:, self.sh.nodes]
        f64 = dict(dtype=torch.float64, device=dev)
        self.U = torch.from_numpy(np.ascontiguousarray(U_h.reshape(-1, order='F'))).to(dev)
        self.Ep = torch.zeros((4, n_int), **f64)
        self.S = torch.empty((4, n_int), **f64)
        self.DS = torch.empty((9, n_int), **f64)
        self.indp = torch.empty(n_int, dtype=torch.uint8, device=dev)
        self.Kd = torch.empty(ctx.nnz, **f64)
        self.Fb = [torch.empty(ctx.n_dof, **f64) for _ in range(2 if two_buffers else 1)]
        self.counts = torch.zeros(2, dtype=torch.int64, device=dev)

    def step(self, stream, i=0, full=True):
        """full: every output of SURVEY 8d (s, ds, ind_p, K, F); else only what a Newton iterate reads (K, F)."""
        if self.idle:
            return
        self.ctx.step_dev(stream, self.U.data_ptr(), ep=self.Ep.data_ptr(), accept=False,
                          s=self.S.data_ptr() if full else 0, ds=self.DS.data_ptr() if full else 0,
                          ind_p=self.indp.data_ptr() if full else 0, k_data=self.Kd.data_ptr(),
                          f_out=self.Fb[i].data_ptr(), counts=self.counts.data_ptr() if full else 0)   # a Newton iterate logs no counts


def run(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
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
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    fep = importlib.import_module('fem-elastoplasticity_amd')
    fep.lib()                                      # fails loudly if the HIP extension is missing
    et = args.elem
    N = args.cells if args.cells is not None else DEFAULT_CELLS[et]
    f64 = dict(dtype=torch.float64, device=dev)
    main = torch.cuda.current_stream()
    stream = main.cuda_stream
    comm = torch.cuda.Stream() if world > 1 else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(shard, steps, warmup, full=True, xmode=args.exchange):
        """warmup untimed + EXACTLY `steps` timed passes between barrier+synchronize; max over ranks.
        N > 1: the interface exchange of pass i (xmode: 'allreduce' | 'p2p'; None = no exchange at all, the reference point
        of the exposed share) runs on a second stream under pass i+1's kernels (the force vector is double-buffered; a pass
        only waits for the exchange that last used its buffer)."""
        ev_done = [torch.cuda.Event() for _ in range(2)]
        ev_ready = [torch.cuda.Event() for _ in range(2)]
        state = {'i': 0}
        xch = world > 1 and xmode is not None

        def step():
            i = state['i'] & 1 if world > 1 else 0
            state['i'] += 1
            if xch and state['i'] > 2:
                main.wait_event(ev_done[i])
            shard.step(stream, i, full)
            if xch:
                ev_ready[i].record(main)
                comm.wait_event(ev_ready[i])
                with torch.cuda.stream(comm):
                    shard.sh.exchange_force_(shard.Fb[i], mode=xmode)
                    ev_done[i].record(comm)

        if args.preheat_ms > 0:                         # clock ramp: not part of W, not timed (reported as `preheat_ms`)
            # LOCAL kernels only: a loop bounded by wall time runs a different number of passes on every rank, so it must not
            # hold a collective (the interface exchange of step() would leave the ranks waiting for each other for good)
            t_h = time.perf_counter()
            while time.perf_counter() - t_h < args.preheat_ms * 1e-3:
                for _ in range(50):
                    shard.step(stream, 0, full)
                torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        barrier()
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev_a.record(main)                               # HIP events on the launch stream around the K timed steps
        for _ in range(steps):
            step()
        ev_b.record(main)
        barrier()
        dt = time.perf_counter() - t0
        state['region_ms'] = ev_a.elapsed_time(ev_b)
        tmax = torch.tensor([dt], **f64)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), step, state['region_ms'] / max(steps, 1)

    def per_kernel(shard, step, steps):
        """per-kernel durations IN SITU: the same step sequence again with HIP events around every kernel on the
        launch stream (not part of the timed region)."""
        shard.ctx.profile_begin()
        for _ in range(steps):
            step()
        kms, n_prof = shard.ctx.profile_end(stream)
        barrier()
        return kms, n_prof

    def exchange_ms(shard, mode, reps=20):
        """One exchange on its own (back to back on the comm stream, nothing to hide under): max over ranks."""
        if world == 1:
            return None
        barrier()
        with torch.cuda.stream(comm):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            shard.sh.exchange_force_(shard.Fb[0], mode=mode)
            a.record(comm)
            for _ in range(reps):
                shard.sh.exchange_force_(shard.Fb[0], mode=mode)
            b.record(comm)
        barrier()
        t = torch.tensor([a.elapsed_time(b) / reps], **f64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def exchange_report(shard, dt_headline):
        """Both forms of the interface exchange in the SAME run: the exchange alone, the step with it on
        the second stream, and the step without any exchange — exposed = what the overlapped exchange adds to a step, hidden =
        the rest of its own duration."""
        if world == 1:
            return None
        other = 'p2p' if args.exchange == 'allreduce' else 'allreduce'
        dt_other, _, _ = timed(shard, args.steps, args.warmup, xmode=other)
        dt_none, _, _ = timed(shard, args.steps, args.warmup, xmode=None)
        step_ms = {args.exchange: dt_headline / args.steps * 1e3, other: dt_other / args.steps * 1e3, 'none': dt_none / args.steps * 1e3}
        alone = {m: exchange_ms(shard, m) for m in ('allreduce', 'p2p')}
        exposed = {m: max(0.0, step_ms[m] - step_ms['none']) for m in ('allreduce', 'p2p')}
        return {'headline': args.exchange, 'alone_ms': alone, 'step_ms': step_ms, 'exposed_ms': exposed,
                'hidden_ms': {m: max(0.0, alone[m] - exposed[m]) for m in alone},
                'bytes_per_rank': {'allreduce': 16 * shard.sh.n_iface, 'p2p': 8 * int(shard.sh.p2p_send_dofs.size)},
                'neighbours_of_rank0': list(shard.sh.p2p_ranks),
                'note': 'alone: back to back on the comm stream; step: exchange of pass i under the kernels of pass i+1 (two '
                        'streams, double-buffered force); exposed = step - step without exchange; max over ranks'}

    def gather_ranks(v):
        t = torch.tensor(v, **f64)
        if world == 1:
            return [t.tolist()]
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.tolist() for o in out]

    # ---- the headline run ------------------------------------------------------------------------------------
    strong = args.scaling == 'strong'
    mesh = fep.rect_mesh(N, N if strong else N * world, et, 10, 10 if strong else 10 * world)
    shard = Shard(fep, torch, mesh, rank, world, dev, args.field_scale, world > 1, args.state, args.exchange,
                  args.min_elements_per_rank)
    n_el_total = int(mesh['elements'].shape[1])
    n_total = n_el_total * NQ[et]                      # integration points = element*quadpt updates per step, all ranks
    dt, step, stream_ms = timed(shard, args.steps, args.warmup)
    cnt = shard.counts.clone()
    if world > 1:
        dist.all_reduce(cnt)
    n_smooth, n_apex = [int(v) for v in cnt.cpu()]
    if shard.idle:
        kms, n_prof = {'element': 0.0, 'csr': 0.0, 'force': 0.0}, 0
        barrier()
    else:
        kms, n_prof = per_kernel(shard, step, args.steps)
    ranks_ms = gather_ranks([kms['element'], kms['csr'], kms['force']])
    x_rep = exchange_report(shard, dt)
    n_int = shard.n_int
    ctx = shard.ctx

    # K/F-only pass (what a Newton iterate asks for: no s / ds / ind_p leave the kernels), N = 1
    kf = None
    if world == 1:
        dt_kf, step_kf, stream_ms_kf = timed(shard, args.steps, args.warmup, full=False)
        kms_kf, _ = per_kernel(shard, step_kf, args.steps)
        # algorithmic bytes of a K,F-only step: every node's coordinates and displacement once (16 + 16 B),
        # the previous plastic strain (32 B per point), the CSR values written once (8 B each), the force (16 B per node);
        # static index data (element table, gather plan) excluded as in SURVEY 8d
        n_nodes = ctx.n_dof // 2
        kf_bytes = 32.0 * n_nodes + 32.0 * n_int + 8.0 * ctx.nnz + 16.0 * n_nodes
        kf_ms = stream_ms_kf
        kf = {'ms_per_step': dt_kf / args.steps * 1e3, 'stream_ms_per_step': stream_ms_kf, 'updates_per_s': n_int * args.steps / dt_kf,
              'kernels_ms': {'point': kms_kf['element'], 'assembly': kms_kf['csr'], 'force': kms_kf['force']},
              'kernel': ctx.kernel_names(1),
              'algorithmic_bytes': kf_bytes, 'GBps': kf_bytes / (kf_ms * 1e-3) / 1e9,
              'frac': kf_bytes / (kf_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
              'algorithmic_bytes_formula': '32*n_nodes (xy + U) + 32*n_int (ep) + 8*nnz (K) + 16*n_nodes (F)',
              'note': 'fep_step_dev with s = ds = ind_p = counts = NULL (newton.py asks for K and F only); frac = algorithmic '
                      'bytes / stream_ms_per_step / 8 TB/s'}

    # the mesh-free entry point on its own (fep_return_map_dev: the drop-in for construct_constitutive_problem, DP:604-757):
    # strain, plastic strain and the four material arrays in, s / ds / ind_p / counts out = 193 B per point
    rm = None
    if world == 1:
        from importlib import import_module
        _lib = import_module('fem-elastoplasticity_amd._lib')
        E = torch.empty((3, n_int), **f64)
        ctx.step_dev(stream, shard.U.data_ptr(), ep=shard.Ep.data_ptr(), e_out=E.data_ptr(), s=shard.S.data_ptr())
        mats = [torch.full((n_int,), v, **f64) for v in dp_materials()]

        def rm_step():
            _lib.check(_lib.lib().fep_return_map_dev(dev.index, stream, n_int, E.data_ptr(), 1, n_int, None, shard.Ep.data_ptr(),
                                                     mats[0].data_ptr(), mats[1].data_ptr(), mats[2].data_ptr(), mats[3].data_ptr(),
                                                     0, shard.S.data_ptr(), shard.DS.data_ptr(), shard.indp.data_ptr(),
                                                     shard.counts.data_ptr()), 'fep_return_map_dev')
        for _ in range(args.warmup):
            rm_step()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(main)
        for _ in range(args.steps):
            rm_step()
        ev1.record(main)
        torch.cuda.synchronize()
        rm_ms = ev0.elapsed_time(ev1) / args.steps
        rm_cnt = [int(v) for v in shard.counts.cpu()]
        rm = {'ms_per_call': rm_ms, 'points_per_s': n_int / (rm_ms * 1e-3), 'bytes_per_point': 193,
              'GBps': 193.0 * n_int / (rm_ms * 1e-3) / 1e9, 'frac_of_peak': 193.0 * n_int / (rm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
              'counts': rm_cnt, 'note': 'fep_return_map_dev (return_map_kernel + counts_reduce_kernel), per-point material arrays'}
        assert rm_cnt == [n_smooth, n_apex], (rm_cnt, n_smooth, n_apex)
        del E, mats

    # strong-scaling companion of a weak run (N > 1): ONE N x N square split over the ranks
    strong_line = None
    if world > 1 and not strong:
        mesh_s = fep.rect_mesh(N, N, et, 10, 10)
        shard_s = Shard(fep, torch, mesh_s, rank, world, dev, args.field_scale, True, args.state, args.exchange,
                        args.min_elements_per_rank)
        dt_s, step_s, _ = timed(shard_s, args.steps, args.warmup)
        if shard_s.idle:
            kms_s = {'element': 0.0, 'csr': 0.0}
            barrier()
        else:
            kms_s, _ = per_kernel(shard_s, step_s, args.steps)
        rk = gather_ranks([kms_s['element'], kms_s['csr'], float(shard_s.n_int)])
        n_s = int(mesh_s['elements'].shape[1]) * NQ[et]
        strong_line = {'value': n_s * args.steps / dt_s, 'unit': 'updates/s', 'ms_per_step': dt_s / args.steps * 1e3,
                       'points_total': n_s, 'elements_total': int(mesh_s['elements'].shape[1]), 'scaling': 'strong',
                       'active_ranks': shard_s.sh.active_world,
                       'per_rank': [{'points': int(r[2]), 'point_ms': r[0], 'assembly_ms': r[1]} for r in rk],
                       'exchange': exchange_report(shard_s, dt_s)}
        shard_s.sh.close()

    if rank != 0:                                   # everything below describes and prints rank 0's line; no collective follows
        dist.barrier()
        dist.destroy_process_group()
        return
    # what the library runs (FEP_ROUTE: unset = the product's default route; coo / patch = the element route's cross-check forms)
    env_route = os.environ.get('FEP_ROUTE', '')
    route = ('node' if not env_route else 'coo' if env_route == 'coo' else 'element') if et == 'P1' else 'element'
    patch_form = env_route != 'coo'
    k_names = ctx.kernel_names(0)                   # as rocprofv3 prints them (fep_ctx_kernel_names)
    if et != 'P1' or route != 'node':
        # element route: strain + return map + K_e blocks in one kernel; patch form (default): closed CSR blocks written by the
        # same kernel, fixup_kernel for the node pairs on patch boundaries — the priced work is the pair
        k_ms = kms['element'] + (kms['csr'] if patch_form else 0.0)
        if patch_form:
            k_name = k_names + ' (strain + return map + K_e in LDS + closed CSR blocks; patch-boundary blocks)'
            others = {'element_kernel': kms['element'], 'fixup_kernel': kms['csr']}
        else:
            k_name = k_names.split(' + ')[0] + ' (strain + return map + K_e/f_e blocks to HBM)'
            others = {'csr_reduce_kernel': kms['csr'], 'force_reduce_kernel': kms['force']}
        per_k = None
    else:
        # node route: the priced work (return map + tangent assembly) is the PAIR of kernels; the
        # figure divides SURVEY 8d's bytes by the SUM of both durations (which also includes the CSR
        # numeric phase and the force gather that 8d prices separately)
        k_ms = kms['element'] + kms['csr']
        k_name = k_names + ' (strain + return map; tangent CSR values + force)'
        n_point, n_asm = k_names.split(' + ')
        others = {n_point: kms['element'], n_asm: kms['csr']}
        # each kernel against its OWN minimal HBM bytes: point = elem ids 12 + ep 32
        # + coordinates/displacements 32 (16 B per node, ~2 elements per node) + s 32 + ds 72 + ind_p 1 = 181 B (the
        # materials are constant over the mesh here and are not read: 32 B less than the general case);
        # assembly = ds 48 (6 of 9 rows) + s 24 + geometry record 48 + descriptors/codes 32 + CSR values 8*nnz/n + force 16*n_n/n
        b_point = 181.0 * n_int
        b_node = (48 + 24 + 48 + 32) * n_int + 8.0 * ctx.nnz + 8.0 * ctx.n_dof
        per_k = {n_point: {'bytes': b_point, 'GBps': b_point / (kms['element'] * 1e-3) / 1e9,
                           'frac': b_point / (kms['element'] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                 n_asm: {'bytes': b_node, 'GBps': b_node / (kms['csr'] * 1e-3) / 1e9,
                         'frac': b_node / (kms['csr'] * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    alg = ALG_BYTES[et] * n_int
    # Duration of the priced kernels per launch: the HIP events around the K TIMED steps on the launch stream, divided by K.  In
    # the default forms a step enqueues the priced kernels and nothing else (P1 node route: point + assembly kernel; patch form:
    # element + fix-up kernel), so this is the sum of their durations plus the gaps between them — an upper bound that agrees with
    # the rocprofv3 kernel stats to a few per cent.  The per-kernel split below comes from a second pass with an event pair around
    # EVERY kernel, which costs 2-5 us per pair and serialises the launches (its sum is 5-12 % above the stream time).
    timing = f'HIP events in situ around every kernel, mean of {n_prof} launches'
    k_ms_split = k_ms
    # N > 1: the launch stream also waits for the interface exchange of two passes earlier (double-buffered force vector), which
    # is not kernel time: the per-kernel event pairs stay the priced duration there.
    if world == 1 and (route == 'node' or patch_form):
        k_ms = stream_ms
        timing = (f'HIP events on the launch stream around the {args.steps} timed steps / {args.steps} (the step\'s kernels back to '
                  f'back); kernels_ms: a second pass with an event pair around every kernel (sum {k_ms_split:.4f} ms)')
    achieved = alg / (k_ms * 1e-3) / 1e9

    if rank == 0:
        # HBM traffic is NOT measured by this run (PMC counters need their own rocprofv3 passes): the figure is the
        # one recorded by tools/summarize_profile.py from the committed counter passes of the same command
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the recorded counter passes must be of the kernels this run is about to price, on this workload — else null
                same_kernels = sorted(tj.get('kernels', {})) == sorted(k_names.split(' + '))
                if (same_kernels and tj.get('elements_per_gpu', 1002528) == n_int // NQ[et] and tj.get('element_type', 'P1') == et
                        and args.field_scale == 1.0 and args.state == tj.get('state', 'bands')):
                    traffic = tj.get('hbm_bytes_per_launch')
                    traffic_source = ('profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                                      '`python bench.py`, 2*FETCH + WRITE per the gfx950 correction; not measured in this run)')
                    kf_names = ctx.kernel_names(1).split(' + ')
                    if kf and sorted(tj.get('kf_only_kernel', {})) == sorted(kf_names):
                        kf['traffic'] = sum(tj['kf_only_kernel'].values())
                elif not same_kernels:
                    print(f'[bench] profiles/traffic_latest.json is of other kernels ({sorted(tj.get("kernels", {}))}): traffic null',
                          file=sys.stderr)
            except Exception:
                traffic = None
        label = ' (BASELINE configs[3])' if (et, N) == ('P1', N_CELLS) else ' (BASELINE configs[4])' if (et, N) == ('P2', 1414) else ''
        pts = '' if et == 'P1' else f' x {NQ[et]} points'
        if strong:
            wl = (f'Plasticity2D_DP strip-footing square, {N}x{N} cells = {n_el_total} {et} elements{pts}{label} split over '
                  f'{world} GPU(s)')
        else:
            wl = (f'Plasticity2D_DP strip-footing square, {N}x{N} cells = {n_int // NQ[et]} {et} elements{pts} per GPU{label}')
        line = {
            'metric': 'element*quadpt updates/sec (return-map + K_tan assemble)',
            'value': n_total * args.steps / dt, 'unit': 'updates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'preheat_ms': args.preheat_ms,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': wl + ', Drucker-Prager, strain->return map->K_tan CSR values->F per step',
                       'element_type': et, 'points_per_gpu': n_int, 'points_total': n_total, 'elements_total': n_el_total,
                       'nnz_per_gpu': ctx.nnz,
                       'smooth_points': n_smooth, 'apex_points': n_apex, 'field_scale': args.field_scale, 'state': args.state,
                       'route': route,
                       'elements_per_gpu': n_int // NQ[et],
                       'parallelism': (f'element-shard x{world} ({shard.sh.active_world} active), interface-force exchange: '
                                       f'{args.exchange}') if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'kernel': k_name, 'kernel_ms': k_ms,
                         'algorithmic_bytes_per_launch': alg, 'timing': timing, 'kernel_ms_event_pairs': k_ms_split,
                         'kernels_ms': others, 'per_kernel': per_k},
        }
        if world > 1:
            line['per_rank'] = [{'point_ms': r[0], 'assembly_ms': r[1]} for r in ranks_ms]      # first / second kernel of the step
            line['exchange'] = x_rep
            line['exchange_ms'] = x_rep['alone_ms']                   # {'allreduce': ms, 'p2p': ms}; the headline used x_rep['headline']
            line['active_ranks'] = shard.sh.active_world               # < n_gpus: the min-elements-per-rank gate left ranks idle
            if strong_line:
                line['strong'] = strong_line
        if kf:
            line['kf_only'] = kf
        if rm:
            line['return_map_only'] = rm
        if world == 1 and not args.no_cpu_baseline:
            cpu_n = args.cpu_cells or (N if et == 'P1' else min(N, 354))
            line['cpu_baseline'], oracle_out = cpu_baseline(fep, et, n_cells=cpu_n, scale=args.field_scale, state=args.state)
            if cpu_n == N:
                # the checker ran the very mesh and field of the GPU step: compare (full-output pass once more, outside timing)
                shard.step(stream, 0, True)
                torch.cuda.synchronize()
                errs = parity_vs_oracle(oracle_out, shard, torch)
                line['parity_vs_oracle'] = errs
                line['parity_max_rel'] = max(errs['s'], errs['ds'], errs['K'], errs['K_rows'], errs['F'])
                if errs['ind_p_mismatches'] or line['parity_max_rel'] > 1e-11:      # reported, never fatal: the line is the product
                    print(f'[bench] parity against the oracle outside the stated tolerance: {errs}', file=sys.stderr)
        print(json.dumps(line), flush=True)
        try:        # the box's clocks next to the line (stderr): runs of the same binary differ by ~10 % between boxes of the pool
            # (a profiler's preloaded library must not ride into the child: with --pmc it starts the GPU runtime in every process it
            # is loaded into, and rocm-smi — an `env python3` script — would then replace a GPU-initialised process, which the box refuses)
            env_smi = {k: v for k, v in os.environ.items() if k != 'LD_PRELOAD' and not k.startswith(('ROCP', 'ROCPROF'))}
            smi = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                 text=True, timeout=20, env=env_smi).stdout
            keep = [l.strip() for l in smi.splitlines() if 'GPU[0]' in l and any(k in l for k in ('sclk', 'mclk', 'fclk', 'Power'))]
            print('[bench] ' + ' | '.join(keep), file=sys.stderr)
        except Exception as exc:                                        # rocm-smi missing or refused: the line stands
            print(f'[bench] rocm-smi not available: {exc}', file=sys.stderr)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    run(args)


if __name__ == '__main__':
    main()
