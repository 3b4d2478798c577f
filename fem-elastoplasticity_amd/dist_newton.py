"""
The Newton / load-step driver on an element-sharded mesh (SURVEY 8e + 8f row 1): one process per GPU, every rank keeps
its part of the iterate, of K and of F on its device.

    K = sum_r P_r^T K_r P_r      sub-assembled: rank r holds K_r on its local nodes (sharding.ShardedContext), rows of
                                 interface nodes are partial sums
    vectors                      CONSISTENT: every rank holds the full value on all its local DOFs (interface DOFs are
                                 duplicated); a product K_r x_r is partial on the interface and becomes consistent by the
                                 interface exchange (the all-reduce of sharding.Partition.exchange_force_)
    inner products               sum over ranks of  sum_i w_i a_i b_i,  w_i = 1 / (number of ranks that hold DOF i)

DistributedPCG: conjugate gradients on K[Q][:,Q] x[Q] = b[Q] (the reference's np.linalg.solve, DP:1062-1066) with the
2x2 node-block Jacobi preconditioner of the single-GPU solver (the diagonal blocks of interface nodes are summed over the
ranks once per solve).  Per iteration: one local block SpMV (fep_solver_spmv_dev on K_r where the assembly kernels left
it), one interface exchange, two small all-reduces (p.Kp; r.z and r.r together).  The loop is driven from the host with
torch tensors as device vectors; convergence is looked at every `check_every` iterations.

solve_strip_footing_sharded: Plasticity2D_DP's load-step loop (newton.solve_strip_footing, DP:986-1131) on the ranks of
the default process group; same step control, same stopping norms (energy norms through the weighted inner product).

The reference has no parallelism of any kind; this module is new.
"""
import numpy as np

from .mesh import square_mesh
from .sharding import ShardedContext
from .tables import _coerce, element_tables


def _allreduce_(t, group=None, device=None):
    """In-place sum over the ranks.  The tensor goes where the backend can reduce it: device tensors through the host when
    the backend is gloo (rehearsal on one GPU), HOST tensors through the rank's device (`device`, default: torch's current
    one) when it is not — an RCCL process group rejects CPU tensors."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    gloo = dist.get_backend(group) == 'gloo'
    if t.is_cuda and gloo:
        tmp = t.cpu()
        dist.all_reduce(tmp, group=group)
        t.copy_(tmp)
    elif not t.is_cuda and not gloo:
        tmp = t.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))
        dist.all_reduce(tmp, group=group)
        t.copy_(tmp)
    else:
        dist.all_reduce(t, group=group)
    return t


class DistributedPCG:
    """Block-Jacobi conjugate gradients on the sub-assembled K of a ShardedContext."""

    def __init__(self, sc, free_dof_global, group=None):
        import torch
        from .solver import KrylovSolver
        self.torch, self.sc, self.group = torch, sc, group
        ctx = sc.ctx
        self.dev = torch.device('cuda', ctx.device)
        dofs = (2 * sc.nodes[:, None] + np.arange(2)[None, :]).ravel()
        self.dofs_global = dofs
        free = np.asarray(free_dof_global).ravel()[dofs] != 0
        self.solver = KrylovSolver(ctx, free)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.free = torch.from_numpy(free.astype(np.float64)).to(self.dev)
        self.w = torch.from_numpy(np.repeat(1.0 / sc.mult, 2)).to(self.dev)          # 1 / multiplicity per local DOF
        self.wfree = self.w * self.free
        # positions of every node's diagonal 2x2 block in the CSR data of the local pattern
        ip, ix = ctx.pattern()
        n_n = ctx.n_n
        rows0 = ip[0:2 * n_n:2].astype(np.int64)
        rows1 = ip[1:2 * n_n:2].astype(np.int64)
        row_of = np.repeat(np.arange(2 * n_n, dtype=np.int64), np.diff(ip.astype(np.int64)))
        hit = np.flatnonzero((row_of % 2 == 0) & (ix == row_of))         # entry (2n, 2n) of every node that has a block
        has = np.zeros(n_n, dtype=bool)
        p = np.zeros(n_n, dtype=np.int64)
        has[row_of[hit] // 2] = True
        p[row_of[hit] // 2] = hit - rows0[row_of[hit] // 2]
        self._d_idx = [torch.from_numpy(a).to(self.dev) for a in (rows0 + p, rows0 + p + 1, rows1 + p, rows1 + p + 1)]
        self._has = torch.from_numpy(has.astype(np.float64)).to(self.dev)
        self.n_dof = ctx.n_dof
        self.tmp = [torch.empty(self.n_dof, **f64) for _ in range(2)]
        self.last = None

    def close(self):
        self.solver.close()

    # ---- building blocks ---------------------------------------------------------------------------------------
    def exchange_(self, v):
        return self.sc.exchange_force_(v, group=self.group)

    def dot(self, a, b):
        return _allreduce_((self.w * a * b).sum().reshape(1), self.group)[0]

    def spmv(self, k_data, x, masked=False):
        y = self.solver.spmv(k_data, x, masked=masked)
        return self.exchange_(y)

    def _block_jacobi(self, k_data):
        """Inverse of the assembled 2x2 diagonal blocks restricted to the free DOFs, as four per-node vectors."""
        t = self.torch
        d = [k_data[i] * self._has for i in self._d_idx]                 # d00, d01, d10, d11 of K_r
        a = t.stack([d[0], d[3]], dim=1).reshape(-1).contiguous()         # (d00, d11) and (d01, d10) as DOF vectors:
        b = t.stack([d[1], d[2]], dim=1).reshape(-1).contiguous()         # summed over the ranks by the interface exchange
        self.exchange_(a)
        self.exchange_(b)
        f0, f1 = self.free[0::2], self.free[1::2]
        d00 = t.where(f0 > 0, a[0::2], t.ones_like(f0))
        d11 = t.where(f1 > 0, a[1::2], t.ones_like(f1))
        both = f0 * f1
        d01, d10 = b[0::2] * both, b[1::2] * both
        d00 = t.where(d00 == 0, t.ones_like(d00), d00)                   # nodes of no element
        d11 = t.where(d11 == 0, t.ones_like(d11), d11)
        det = d00 * d11 - d01 * d10
        return d11 / det, -d01 / det, -d10 / det, d00 / det

    def _apply_m(self, Mi, r):
        t = self.torch
        r0, r1 = r[0::2], r[1::2]
        z = t.stack([Mi[0] * r0 + Mi[1] * r1, Mi[2] * r0 + Mi[3] * r1], dim=1).reshape(-1)
        return z * self.free

    def pcg(self, k_data, b, rtol=1e-11, max_iter=100000, check_every=10):
        """x (consistent, 0 on constrained DOFs) with |r| <= rtol |b[Q]| (weighted norms = the global Euclidean ones)."""
        t = self.torch
        Mi = self._block_jacobi(k_data)
        r = (b * self.free).clone()
        x = t.zeros_like(r)
        z = self._apply_m(Mi, r)
        p = z.clone()
        s = _allreduce_(t.stack([(self.w * r * z).sum(), (self.w * r * r).sum()]), self.group)
        rz, bb = s[0], float(s[1])
        if bb == 0.0:
            self.last = {'iters': 0, 'relres': 0.0, 'state': 1}
            return x
        state, it, rr = 0, 0, bb
        while it < max_iter:
            q = self.spmv(k_data, p, masked=True)
            pq = self.dot(p, q)
            alpha = rz / pq
            x.add_(alpha * p)
            r.sub_(alpha * q)
            z = self._apply_m(Mi, r)
            s = _allreduce_(t.stack([(self.w * r * z).sum(), (self.w * r * r).sum()]), self.group)
            beta = s[0] / rz
            rz = s[0]
            p = z + beta * p
            it += 1
            if it % check_every == 0 or it == max_iter:
                rr = float(s[1])
                pqh = float(pq)
                if not np.isfinite(rr) or not pqh > 0.0:
                    state = 2
                    break
                if rr <= (rtol ** 2) * bb:
                    state = 1
                    break
        self.last = {'iters': it, 'relres': float(np.sqrt(rr / bb)), 'state': state}
        return x


class _ShardOps:
    """newton._DeviceOps on a ShardedContext: local vectors (consistent on the interface), sub-assembled K."""

    def __init__(self, sc, qf_global, rtol=1e-11, max_iter=200000, inexact_rtol=None, group=None):
        import torch
        self.torch, self.sc, self.ctx, self.group = torch, sc, sc.ctx, group
        self.dev = torch.device('cuda', self.ctx.device)
        self.cg = DistributedPCG(sc, qf_global, group)
        self.rtol, self.max_iter, self.inexact_rtol = rtol, max_iter, inexact_rtol
        f64 = dict(dtype=torch.float64, device=self.dev)
        ctx = self.ctx
        self.kd = torch.empty(ctx.nnz, **f64)
        self.F = torch.empty(ctx.n_dof, **f64)
        self.s = torch.empty((4, ctx.n_int), **f64)
        self.ind = torch.empty(ctx.n_int, dtype=torch.uint8, device=self.dev)
        self.counts = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.pcg_iters = []
        self.n_dof_global = int(np.asarray(qf_global).size)

    def vec(self, a_global):
        a = np.asarray(a_global, dtype=np.float64).ravel()[self.cg.dofs_global]
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def zeros(self):
        return self.torch.zeros(self.ctx.n_dof, dtype=self.torch.float64, device=self.dev)

    def new_ep(self):
        return self.torch.zeros((4, self.ctx.n_int), dtype=self.torch.float64, device=self.dev)

    def step(self, U, Ep=None, accept=False, e0=None, want=('K', 'F'), keep_K=False):
        t = self.torch
        st = t.cuda.current_stream(self.dev).cuda_stream
        kd = None
        if 'K' in want:
            kd = t.empty(self.ctx.nnz, dtype=t.float64, device=self.dev) if keep_K else self.kd
        logs = 's' in want or 'ind_p' in want
        self.ctx.step_dev(st, U.data_ptr(), ep=0 if Ep is None else Ep.data_ptr(), accept=accept and Ep is not None, e0=e0,
                          s=self.s.data_ptr() if 's' in want else 0, ind_p=self.ind.data_ptr() if 'ind_p' in want else 0,
                          k_data=0 if kd is None else kd.data_ptr(), f_out=self.F.data_ptr() if 'F' in want else 0,
                          counts=self.counts.data_ptr() if logs else 0)
        if 'F' in want:
            self.sc.exchange_force_(self.F, group=self.group)            # the one exchange of the hot path: interface forces
        out = {k: v for k, v in {'K': kd, 'F': self.F, 's': self.s, 'ind_p': self.ind}.items() if k in want}
        if logs:
            c = _allreduce_(self.counts.clone(), self.group).cpu()
            out['n_smooth'], out['n_apex'] = int(c[0]), int(c[1])
        return out

    def solve(self, K, rhs, criterion=None):
        rtol = self.rtol
        if self.inexact_rtol and criterion is not None:
            rtol = max(self.rtol, self.inexact_rtol)
        x = self.cg.pcg(K, rhs, rtol=rtol, max_iter=self.max_iter)
        self.pcg_iters.append(self.cg.last['iters'])
        if self.cg.last['state'] != 1:
            x.fill_(float('nan'))
        return x

    def matvec(self, K, v):
        return self.cg.spmv(K, v)

    def energy(self, K, v):
        return float(self.torch.sqrt(self.cg.dot(v, self.cg.spmv(K, v))))

    def host(self, v):
        """The GLOBAL vector on every rank (every DOF from the ranks that hold it, weighted: they agree)."""
        t = self.torch
        if v.dim() == 2:                                                 # point data (rows, n_int): this rank's slice only
            return v.cpu().numpy()
        g = t.zeros(self.n_dof_global, dtype=t.float64)
        g[t.from_numpy(self.cg.dofs_global)] = (v * self.cg.w).cpu()
        return _allreduce_(g, self.group, self.dev).numpy()

    def nodal(self, q_int, elem_global, weight_local):
        """transform (DP:760-816) of a point field: numerators and denominators summed over the ranks, global nodal array."""
        t = self.torch
        le = self.sc.local_elements
        n_p, n_e = le.shape
        w = np.asarray(weight_local, dtype=float).ravel()
        n_q = w.size // n_e
        nodes = self.sc.nodes[np.repeat(le, n_q, axis=1)]
        n_n = self.n_dof_global // 2
        wq = w * q_int.cpu().numpy().ravel()
        f = t.from_numpy(np.stack([np.bincount(nodes.ravel(), weights=np.tile(wq, n_p), minlength=n_n),
                                   np.bincount(nodes.ravel(), weights=np.tile(w, n_p), minlength=n_n)]))
        _allreduce_(f, self.group, self.dev)
        return (f[0] / f[1]).numpy()

    def close(self):
        self.cg.close()


def solve_strip_footing_sharded(element_type='P1', level=1, n_cells=None, size_xy=10, max_steps=None, zeta_max=1.0,
                                device=None, log=None, pcg_rtol=1e-11, pcg_inexact_rtol=None, keep_U=True, group=None):
    """newton.solve_strip_footing on the ranks of the process group (torch.distributed initialised by the caller; a
    single process works too): the mesh is split by contiguous element ranges, every rank runs the hot path on its
    shard and the distributed conjugate gradients above solve the Newton corrections.  Returns the same history on every
    rank ('U' holds GLOBAL displacement fields)."""
    import torch.distributed as dist
    from . import newton
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    t = _coerce(element_type)
    mesh = square_mesh(size_xy * 2 ** level if n_cells is None else n_cells, t, size_xy)
    d1, d2, wf = element_tables(t)
    holder = {}

    def factory(elem, coord, dh1, dh2, w):
        holder['sc'] = ShardedContext(elem, coord, rank, world, dh1, dh2, w, device=device)
        return holder['sc'].ctx

    def make_ops(ctx, qf, *a, **k):
        return _ShardOps(holder['sc'], qf, rtol=pcg_rtol, inexact_rtol=pcg_inexact_rtol, group=group)
    return newton.solve_strip_footing(element_type, level, n_cells, size_xy, max_steps, zeta_max, device, log, factory,
                                      'pcg', pcg_rtol, keep_U, None, 1e-4, pcg_inexact_rtol, _ops_factory=make_ops)
