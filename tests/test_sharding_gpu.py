"""Element sharding on the GPU, in one process: the shards of a world of 2/3 are built and stepped one after
the other on cuda:0, the interface forces are summed the way the all-reduce does, and everything is compared
with the single-context result on the global mesh.  (The multi-process exchange itself is covered on CPU by
tests/test_sharding_gloo.py.)"""
import os

import numpy as np
import pytest

from conftest import dp_materials, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('t,nx,ny,world', [('P1', 40, 60, 2), ('P1', 30, 45, 3), ('Q1', 24, 36, 2), ('P2', 20, 30, 2), ('P2', 14, 21, 3)])
def test_shards_reproduce_global_step(fep, t, nx, ny, world):
    mesh = fep.rect_mesh(nx, ny, t, 10, 15)
    elem, coord = mesh['elements'], mesh['coordinates']
    n_q = fep.ELEMENT_SHAPE[fep.LagrangeElementType[t]][1]
    n_int = elem.shape[1] * n_q
    sh, bu, eta, c = dp_materials(n_int)
    x, y = coord
    U = np.array([2.5e-4 * y * (x / 10) + 1.2e-4 * x * (y > 5), -1.5e-4 * y * (x < 5) + 2.0e-4 * y * (x >= 5)])
    ctx = fep.MeshContext(elem, coord)
    ctx.set_materials(sh, bu, eta, c)
    ref = ctx.step(U, np.zeros((4, n_int)), want=('s', 'ds', 'ind_p', 'K', 'F'))
    assert ref['n_smooth'] > 0 and ref['n_apex'] > 0
    n_dof = 2 * coord.shape[1]
    F_sum = np.zeros(n_dof)
    K_sum = None
    shards = []
    for r in range(world):
        sc = fep.ShardedContext(elem, coord, r, world)
        sc.set_materials(sh, bu, eta, c)
        out = sc.ctx.step(U[:, sc.nodes], np.zeros((4, sc.ctx.n_int)), want=('s', 'ds', 'ind_p', 'K', 'F'))
        sl = sc.local_point_slice()
        # integration-point data is owned by exactly one rank and equals the global result
        assert np.array_equal(out['ind_p'], ref['ind_p'][sl])
        assert relerr(out['s'], ref['s'][:, sl]) <= 1e-13 and relerr(out['ds'], ref['ds'][:, sl]) <= 1e-13
        dofs = (2 * sc.nodes[:, None] + np.arange(2)[None, :]).ravel()
        F_sum[dofs] += out['F']
        Kg = out['K'].tocoo()
        import scipy.sparse as ssp
        Kr = ssp.coo_matrix((Kg.data, (dofs[Kg.row], dofs[Kg.col])), shape=(n_dof, n_dof)).tocsr()
        K_sum = Kr if K_sum is None else K_sum + Kr
        shards.append((sc, out['F'], dofs))
    assert relerr(F_sum, ref['F']) <= 1e-12                         # partial forces sum to the global force
    assert abs(K_sum - ref['K']).max() <= 1e-12 * np.abs(ref['K'].data).max()    # sub-assembled K
    # what the all-reduce leaves on every rank: summed interface DOFs
    buf = np.zeros(2 * shards[0][0].n_iface)
    for sc, F, _ in shards:
        buf[sc.iface_slot_dofs] += F[sc.iface_local_dofs]
    for sc, F, dofs in shards:
        F = F.copy()
        F[sc.iface_local_dofs] = buf[sc.iface_slot_dofs]
        assert relerr(F, ref['F'][dofs]) <= 1e-12
        sc.close()


def test_step_dev_is_graph_capturable_and_device_resident(fep):
    """fep_step_dev on caller-owned device buffers (torch tensors): captured into a HIP graph, replayed, and equal
    bit for bit to the host-array entry point."""
    import torch
    mesh = fep.square_mesh(40, 'P1', 10)
    elem, coord = mesh['elements'], mesh['coordinates']
    n = elem.shape[1]
    sh, bu, eta, c = dp_materials(n)
    x, y = coord
    U = np.array([2.5e-4 * y * (x / 10) + 1.2e-4 * x * (y > 5), -1.5e-4 * y * (x < 5) + 2.0e-4 * y * (x >= 5)])
    ctx = fep.MeshContext(elem, coord)
    ctx.set_materials(sh, bu, eta, c)
    ref = ctx.step(U, np.zeros((4, n)), want=('s', 'ds', 'ind_p', 'K', 'F'))
    dev = torch.device('cuda', 0)
    f64 = dict(dtype=torch.float64, device=dev)
    Ud = torch.from_numpy(np.ascontiguousarray(U.reshape(-1, order='F'))).to(dev)
    Ep = torch.zeros((4, n), **f64); S = torch.zeros((4, n), **f64); DS = torch.zeros((9, n), **f64)
    ind = torch.zeros(n, dtype=torch.uint8, device=dev); Kd = torch.zeros(ctx.nnz, **f64); F = torch.zeros(ctx.n_dof, **f64)
    cnt = torch.zeros(2, dtype=torch.int64, device=dev)

    def launch():
        ctx.step_dev(torch.cuda.current_stream().cuda_stream, Ud.data_ptr(), ep=Ep.data_ptr(), s=S.data_ptr(),
                     ds=DS.data_ptr(), ind_p=ind.data_ptr(), k_data=Kd.data_ptr(), f_out=F.data_ptr(), counts=cnt.data_ptr())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                                  # warm-up outside capture (lazy allocations)
        launch()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for t in (S, DS, Kd, F):
        t.zero_()
    cnt.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(S.cpu().numpy(), ref['s']) and np.array_equal(DS.cpu().numpy(), ref['ds'])
    assert np.array_equal(Kd.cpu().numpy(), ref['K'].data) and np.array_equal(F.cpu().numpy(), ref['F'])
    assert np.array_equal(ind.cpu().numpy().astype(bool), ref['ind_p'])
    assert tuple(cnt.cpu().tolist()) == (ref['n_smooth'], ref['n_apex'])
    ctx.close()


def test_step_dev_refuses_to_allocate_inside_a_stream_capture(fep):
    """An accepting call that asks for K without ds needs the internal ds scratch of the two-kernel route.  On a fresh P1
    context that scratch does not exist yet: inside a stream capture the call is refused with FEP_ESTATE (no hipMalloc
    under capture, the capture stays intact); outside it allocates, and the same call then captures and replays."""
    import torch
    mesh = fep.square_mesh(24, 'P1', 10)
    elem, coord = mesh['elements'], mesh['coordinates']
    n = elem.shape[1]
    x, y = coord
    U = np.array([2.5e-4 * y * (x / 10) + 1.2e-4 * x * (y > 5), -1.5e-4 * y * (x < 5) + 2.0e-4 * y * (x >= 5)])
    ctx = fep.MeshContext(elem, coord)
    ctx.set_materials(*dp_materials(n))
    dev = torch.device('cuda', 0)
    f64 = dict(dtype=torch.float64, device=dev)
    Ud = torch.from_numpy(np.ascontiguousarray(U.reshape(-1, order='F'))).to(dev)
    Ep = torch.zeros((4, n), **f64); Kd = torch.zeros(ctx.nnz, **f64); F = torch.zeros(ctx.n_dof, **f64)

    def launch():
        ctx.step_dev(torch.cuda.current_stream().cuda_stream, Ud.data_ptr(), ep=Ep.data_ptr(), accept=True,
                     k_data=Kd.data_ptr(), f_out=F.data_ptr())
    g = torch.cuda.CUDAGraph()
    refused = None
    with torch.cuda.graph(g):
        try:
            launch()
        except fep.FepError as e:
            refused = e.code
    assert refused == -6                                           # FEP_ESTATE
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        launch()                                                   # outside a capture: allocates the scratch
    torch.cuda.synchronize()
    K1 = Kd.cpu().numpy().copy()
    Ep.zero_()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        launch()
    Kd.zero_(); Ep.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(Kd.cpu().numpy(), K1) and np.abs(K1).max() > 0
    ctx.close()


def test_pinned_blocks_double_free_and_trim(fep):
    """fep_host_free of a pointer that is already back in the cache is refused (it used to be queued twice and handed to two
    arrays); fep_host_trim gives the idle blocks back."""
    import ctypes
    l = fep.lib()
    p = ctypes.c_void_p()
    assert l.fep_host_alloc(ctypes.byref(p), 1 << 16) == 0 and p.value
    assert l.fep_host_free(p) == 0
    assert l.fep_host_free(p) == -1                                # FEP_EINVAL: released twice
    q = ctypes.c_void_p()
    assert l.fep_host_alloc(ctypes.byref(q), 1 << 16) == 0 and q.value == p.value     # the idle block again
    assert l.fep_host_free(q) == 0 and l.fep_host_trim() == 0
    assert l.fep_host_free(q) == -1                                # no longer a block of the cache
    a = fep._lib.pinned_empty((3, 1000))
    a[...] = 1.5
    assert a.sum() == 4500.0


def test_contexts_and_solvers_release_their_device_memory(fep):
    """Create / destroy cycles (all routes' tables, solver hierarchy) leave the device allocation where it was."""
    import torch
    torch.cuda.synchronize()

    def used():
        free, total = torch.cuda.mem_get_info(0)
        return total - free

    def cycle():
        for et, n in (('P1', 64), ('P2', 24), ('Q1', 40)):
            mesh = fep.square_mesh(n, et, 10)
            ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
            ctx.set_materials(*[v[0] for v in dp_materials(1)])
            K = ctx.step(np.zeros(ctx.n_dof), want=('K',))['K']
            sol = fep.KrylovSolver(ctx, mesh['Q'].flatten(order='F'))
            sol.setup_amg(K, mesh['coordinates'], coarse_nodes=20)
            sol.solve_host(K, np.ones(ctx.n_dof), rtol=1e-8)
            sol.close()
            ctx.close()

    cycle()
    torch.cuda.empty_cache()
    base = used()
    for _ in range(5):
        cycle()
    torch.cuda.empty_cache()
    assert used() - base <= 64 << 20         # allocator granularity, not a per-cycle leak (five cycles allocate ~0.5 GB)


@pytest.mark.parametrize('t,world,exchange', [('P1', 2, 'allreduce'), ('Q1', 2, 'allreduce'), ('P2', 2, 'allreduce'),
                                              ('P1', 3, 'p2p'), ('P2', 2, 'p2p')])
def test_two_process_exchange_on_one_gpu(fep, tmp_path, t, world, exchange):
    """The product's multi-GPU path with real processes: `world` fresh processes share cuda:0 (gloo rendezvous), each
    runs ShardedContext.step_dev + exchange_force_ (pack kernel -> all-reduce -> unpack kernel) on two streams with a
    double-buffered force vector — the sequence bench.py --gpus N drives (tests/shard_worker.py).  After the exchange
    every rank must hold the single-context force on ALL its local DOFs, in both buffers, and the sub-assembled K_r
    must sum to the global K."""
    import socket
    import subprocess
    import sys
    import scipy.sparse as ssp
    import shard_worker as sw
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', FEP_EXCHANGE=exchange)   # p2p: the neighbour-only form of the exchange
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shard_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path), t], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    # the single-context result, meanwhile, in this process
    elem, coord, U = sw.problem(fep, t)
    ctx = fep.MeshContext(elem, coord)
    ctx.set_materials(*sw.materials())
    ref = ctx.step(U, np.zeros((4, ctx.n_int)), want=('K', 'F'))
    assert ref['n_smooth'] > 0 and ref['n_apex'] > 0
    n_dof = ctx.n_dof
    ctx.close()
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    fmax = np.abs(ref['F']).max()
    K_sum, counts, covered = None, np.zeros(2, dtype=np.int64), np.zeros(n_dof // 2, dtype=int)
    n_ifaces = []
    for r in range(world):
        d = np.load(tmp_path / f'rank{r}.npz')
        dofs = (2 * d['nodes'][:, None] + np.arange(2)[None, :]).ravel()
        for key in ('F0', 'F1'):
            assert np.isfinite(d[key]).all()
            assert np.abs(d[key] - ref['F'][dofs]).max() <= 1e-12 * fmax, (r, key)
        Kr = ssp.csr_matrix((d['k_data'], d['indices'], d['indptr']), shape=(dofs.size, dofs.size)).tocoo()
        Kg = ssp.coo_matrix((Kr.data, (dofs[Kr.row], dofs[Kr.col])), shape=(n_dof, n_dof)).tocsr()
        K_sum = Kg if K_sum is None else K_sum + Kg
        counts += d['counts']
        covered[d['nodes']] += 1
        n_ifaces.append(int(d['n_iface']))
    assert all(v == (covered > 1).sum() for v in n_ifaces)             # every rank knows the whole interface
    assert (covered >= 1).all() and (covered > 1).sum() == (81 if t == 'P2' else 41) * (world - 1)       # one node row per cut
    assert tuple(counts) == (ref['n_smooth'], ref['n_apex'])
    assert abs(K_sum - ref['K']).max() <= 1e-12 * np.abs(ref['K'].data).max()


@pytest.mark.gpu
def test_host_tensors_never_reach_an_rccl_all_reduce(monkeypatch):
    """dist_newton._allreduce_ under a backend that is not gloo (RCCL rejects CPU tensors): host tensors — the global
    vectors of _ShardOps.host() / .nodal() — are reduced on the rank's device and copied back; device tensors go straight in."""
    import importlib
    import torch
    import torch.distributed as dist
    dn = importlib.import_module('fem-elastoplasticity_amd.dist_newton')
    seen = []

    def fake_all_reduce(t, op=None, group=None):
        seen.append(t.device.type)
        t.mul_(2.0)                                                   # "two ranks with equal contributions"
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda group=None: 2)
    monkeypatch.setattr(dist, 'get_backend', lambda group=None: 'nccl')
    monkeypatch.setattr(dist, 'all_reduce', fake_all_reduce)
    h = torch.arange(6, dtype=torch.float64)
    out = dn._allreduce_(h, device=torch.device('cuda', 0))
    assert out is h and not h.is_cuda and torch.equal(h, 2.0 * torch.arange(6, dtype=torch.float64))
    d = torch.ones(4, dtype=torch.float64, device='cuda:0')
    dn._allreduce_(d)
    assert torch.equal(d.cpu(), torch.full((4,), 2.0, dtype=torch.float64))
    assert seen == ['cuda', 'cuda']
    monkeypatch.setattr(dist, 'get_backend', lambda group=None: 'gloo')
    seen.clear()
    dn._allreduce_(d)                                                 # gloo: device tensors through the host
    dn._allreduce_(h)
    assert seen == ['cpu', 'cpu']


def test_iface_sum_kernel_is_the_host_rank_order_sum(fep):
    """fep_iface_sum_f64 (last step of the neighbour-only exchange) against Partition._iface_sum_host on the index tables of
    a real partition: a rank of a world of 3 on a mesh whose cuts meet (nodes with two AND with three holders), random local
    force and random received values; bit for bit (0 + c_a + c_b + ... in ascending rank order on both sides)."""
    import torch
    from importlib import import_module
    _lib = import_module('fem-elastoplasticity_amd._lib')
    mesh = fep.rect_mesh(9, 7, 'P1', 10, 10)
    rng = np.random.default_rng(11)
    elem = mesh['elements'][:, rng.permutation(mesh['elements'].shape[1])]       # scattered ranges: many shared nodes, 3 holders too
    n_n = mesh['coordinates'].shape[1]
    part = fep.Partition(elem, n_n, 1, 3, exchange='p2p')
    assert part.p2p_ranks == [0, 2] and (part.mult == 3).any() and (part.mult == 2).any()
    F = rng.normal(size=2 * part.nodes.size)
    recv = rng.normal(size=part.p2p_send_dofs.size)
    want = F.copy()
    part._iface_sum_host(want, recv)
    dev = torch.device('cuda', 0)
    Fd = torch.from_numpy(F.copy()).to(dev)
    Rd = torch.from_numpy(recv).to(dev)
    loc = torch.from_numpy(part.iface_local_dofs.astype(np.int32)).to(dev)
    ptr = torch.from_numpy(part.p2p_ptr).to(dev)
    src = torch.from_numpy(part.p2p_src).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().fep_iface_sum_f64(0, st, loc.numel(), loc.data_ptr(), ptr.data_ptr(), src.data_ptr(), Rd.data_ptr(),
                                            Fd.data_ptr()), 'fep_iface_sum_f64')
    got = Fd.cpu().numpy()
    assert np.array_equal(got, want)
    inner = np.ones(F.size, dtype=bool)
    inner[part.iface_local_dofs] = False
    assert np.array_equal(got[inner], F[inner])                                 # DOFs off the interface are not touched
    assert _lib.lib().fep_iface_sum_f64(0, st, 0, None, None, None, None, None) == 0          # n = 0: nothing to do
    assert _lib.lib().fep_iface_sum_f64(0, st, 4, None, ptr.data_ptr(), src.data_ptr(), Rd.data_ptr(), Fd.data_ptr()) == -1


def test_gated_shards_on_the_device(fep):
    """min_elements_per_rank on ShardedContext: the ranks that take elements reproduce the global step between them, a rank
    left idle holds no context and its exchange calls return at once."""
    import torch
    mesh = fep.rect_mesh(20, 30, 'P1', 10, 15)                                 # 1 200 elements
    elem, coord = mesh['elements'], mesh['coordinates']
    with pytest.warns(UserWarning):
        shards = [fep.ShardedContext(elem, coord, r, 4, min_elements_per_rank=500) for r in range(4)]     # 1200 // 500 = 2 ranks
    assert [s.active for s in shards] == [True, True, False, False] and all(s.gated and s.active_world == 2 for s in shards)
    assert shards[2].ctx is None and shards[3].ctx is None and shards[0].n_iface == 21
    plain = [fep.ShardedContext(elem, coord, r, 2) for r in range(2)]
    for a, b in zip(shards[:2], plain):
        assert np.array_equal(a.nodes, b.nodes) and np.array_equal(a.local_elements, b.local_elements) and (a.lo, a.hi) == (b.lo, b.hi)
    idle = shards[3]
    F = torch.zeros(0, dtype=torch.float64, device='cuda:0')
    assert idle.exchange_force_(F, mode='p2p') is F                             # no neighbours: nothing is sent
    idle.set_materials(1.0, 1.0, 1.0, 1.0)                                      # (no context: a no-op)
    assert idle.local_point_slice() == slice(0, 0)
    for s in shards + plain:
        s.close()


@pytest.mark.parametrize('t,n,pattern', [('P1', 20, 'p1_point_kernel + p1_node_lds_kernel<256, '), ('P2', 10, 'element_kernel<6, 7, true, true, true, 512, 1> + fixup_kernel'),
                                         ('Q1', 12, 'element_kernel<4, 4, true, true, true, 256, 1> + fixup_kernel'),
                                         ('Q2', 8, 'element_kernel<8, 9, true, true, true, 256, 1> + fixup_kernel')])
def test_kernel_names_are_what_the_profiler_prints(fep, t, n, pattern):
    """fep_ctx_kernel_names: the names bench.py labels its roofline with and checks profiles/traffic_latest.json against
    (the committed rocprofv3 summaries profiles/r04_*_kernel_stats.csv carry exactly these strings)."""
    mesh = fep.square_mesh(n, t, 10)
    ctx = fep.MeshContext(mesh['elements'], mesh['coordinates'])
    full, kf = ctx.kernel_names(0), ctx.kernel_names(1)
    assert full.startswith(pattern), full
    if t == 'P1':
        assert kf.startswith('p1_fused_kernel<false, 256, ') and kf.endswith(', 1, 1, false, false>'), kf
        import csv, os
        from conftest import ROOT
        stats = {r['Name'] for r in csv.DictReader(open(os.path.join(ROOT, 'profiles', 'r04_p1_kernel_stats.csv')))}
        if ctx.kernel_names(0).endswith('<256, true, 1, true>'):                # (a structured mesh: the run-table form, as bench.py's)
            assert set(full.split(' + ')) <= stats and kf in stats
    else:
        assert kf == full
    assert not fep.lib().fep_build_is_ablation() or os.environ.get('FEP_LIB_PATH')        # the default library is the product build
    ctx.close()
