"""
Multi-GPU element sharding: one process per GPU, each owns a contiguous range of
elements and runs the hot path on it with no data-path collective; the only exchange is the sum of
the internal-force contributions on nodes shared between ranks (an RCCL all-reduce over the small
interface vector, `torch.distributed`).  K stays sub-assembled: K = sum_r P_r^T K_r P_r, each rank
holding K_r on its local nodes (rows of interface nodes are partial sums) — what a distributed
Krylov solver consumes with the same interface exchange on its products.

`FEP_EXCHANGE=p2p` selects the neighbour-only form of that exchange (sends / receives between the ranks on the two sides
of a cut, `Partition._exchange_p2p`); the all-reduce is the default.

The reference has no parallelism of any kind; this module is new.
"""
import numpy as np

from .hotpath import MeshContext


def element_ranges(n_e, world):
    """Contiguous, balanced element ranges [lo, hi) per rank."""
    return [(n_e * r // world, n_e * (r + 1) // world) for r in range(world)]


class Partition:
    """Host-only part of the sharding: local element range, local node numbering and the
    interface maps of rank `rank` (no GPU needed).

    global node id of local node i: `nodes[i]`.
    `iface_local` : local node ids that other ranks also touch,
    `iface_slot`  : their positions in the global, sorted list of all interface nodes
                    (the layout of the exchanged vector, 2 DOFs per slot),
    `mult`        : per local node the number of ranks that hold it (weights 1/mult make inner products of
                    interface-consistent vectors global ones: dist_newton.py)."""

    def __init__(self, elements, n_n, rank, world):
        elements = np.asarray(elements)
        self.rank, self.world = rank, world
        self.ranges = element_ranges(elements.shape[1], world)
        touch = np.zeros(n_n, dtype=np.int32)
        mine = None
        per_rank = []
        for r, (lo, hi) in enumerate(self.ranges):
            nodes_r = np.unique(elements[:, lo:hi])
            touch[nodes_r] += 1
            per_rank.append(nodes_r)
            if r == rank:
                mine = nodes_r
        self.lo, self.hi = self.ranges[rank]
        self.nodes = mine                                          # sorted global ids of local nodes
        self.local_elements = np.searchsorted(mine, elements[:, self.lo:self.hi])
        iface_global = np.flatnonzero(touch > 1)
        is_iface = touch[mine] > 1
        self.mult = touch[mine].astype(np.float64)                 # number of ranks that hold each local node
        self.n_iface = int(iface_global.size)
        self.iface_local = np.flatnonzero(is_iface)
        self.iface_slot = np.searchsorted(iface_global, mine[is_iface])
        # DOF-level index vectors (DOF = 2*node + comp)
        self.iface_local_dofs = (2 * self.iface_local[:, None] + np.arange(2)[None, :]).ravel()
        self.iface_slot_dofs = (2 * self.iface_slot[:, None] + np.arange(2)[None, :]).ravel()
        self._t = None
        # neighbour lists for the point-to-point form of the exchange: per rank that shares nodes with this one, the LOCAL DOFs of
        # the shared nodes in ascending global node order (the same order on both sides of a cut)
        self.neighbours = {}
        for r, nodes_r in enumerate(per_rank):
            if r == rank:
                continue
            shared = np.intersect1d(mine, nodes_r, assume_unique=True)
            if shared.size:
                loc = np.searchsorted(mine, shared)
                self.neighbours[r] = (2 * loc[:, None] + np.arange(2)[None, :]).ravel()
        self._p2p = None

    # ---- host-array exchange (NumPy; used by the gloo tests and small drivers) -----------------
    def exchange_force_host(self, F_local, group=None):
        import torch
        import torch.distributed as dist
        if self.world > 1 and self._use_p2p():
            t = torch.from_numpy(F_local)                             # (shares memory: updated in place)
            self._exchange_p2p(t, group)
            return F_local
        buf = torch.zeros(2 * self.n_iface, dtype=torch.float64)
        buf[torch.from_numpy(self.iface_slot_dofs)] = torch.from_numpy(F_local[self.iface_local_dofs])
        if self.world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        F_local[self.iface_local_dofs] = buf[torch.from_numpy(self.iface_slot_dofs)].numpy()
        return F_local

    # ---- neighbour-only exchange (FEP_EXCHANGE=p2p): every cut is shared by the two ranks on its sides, so each rank sends its
    # partial sums on the shared DOFs to exactly those ranks and receives theirs — bytes per rank independent of the world size,
    # no collective over all ranks.  Contributions are added in ascending rank order (own included) on every rank that holds a
    # DOF: the same bits everywhere, as the all-reduce gives.  One batch of sends / receives (ncclGroupStart / End under RCCL).
    def _exchange_p2p(self, F_t, group=None, stage_host=False):
        import torch
        import torch.distributed as dist
        if not self.neighbours:
            return F_t
        dev = F_t.device
        if self._p2p is None or self._p2p[0] != dev:
            idx = {r: torch.from_numpy(d.astype(np.int64)).to(dev) for r, d in self.neighbours.items()}
            iface = torch.from_numpy(self.iface_local_dofs.astype(np.int64)).to(dev)
            self._p2p = (dev, idx, iface)
        _, idx, iface = self._p2p
        xdev = torch.device('cpu') if stage_host else dev
        send = {r: F_t[i].to(xdev).contiguous() for r, i in idx.items()}
        recv = {r: torch.empty_like(send[r]) for r in idx}
        ops = []
        for r in sorted(idx):
            ops.append(dist.P2POp(dist.isend, send[r], r, group))
            ops.append(dist.P2POp(dist.irecv, recv[r], r, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        own = F_t[iface]                                             # this rank's partial sums on its interface DOFs (a copy)
        F_t[iface] = 0.0
        for r in sorted(list(idx) + [self.rank]):                    # 0 + c_a + c_b + ...: ascending rank order on every holder
            if r == self.rank:
                F_t[iface] += own
            else:
                F_t[idx[r]] += recv[r].to(dev)
        return F_t

    def _use_p2p(self):
        import os
        return os.environ.get('FEP_EXCHANGE', 'allreduce') == 'p2p'

    # ---- device-resident exchange (torch tensors on the rank's GPU; RCCL) --------------------------
    def exchange_force_(self, F_local_t, group=None):
        """In-place: interface DOFs of the local force tensor become the sum over all ranks.
        pack kernel (other ranks' slots written as zero) -> all-reduce -> unpack kernel, all three on torch's
        CURRENT stream of the tensor's device (`with torch.cuda.stream(s):` selects another one; the collective
        follows torch's current stream, so the kernels must too)."""
        import torch
        import torch.distributed as dist
        from . import _lib
        dev = F_local_t.device
        if self.world > 1 and self._use_p2p():
            return self._exchange_p2p(F_local_t, group, stage_host=dist.get_backend(group) == 'gloo')
        if self._t is None:
            pack = np.full(2 * self.n_iface, -1, dtype=np.int32)
            pack[self.iface_slot_dofs] = self.iface_local_dofs
            self._t = (torch.from_numpy(pack).to(dev), torch.from_numpy(self.iface_slot_dofs.astype(np.int32)).to(dev),
                       torch.from_numpy(self.iface_local_dofs.astype(np.int32)).to(dev),
                       torch.empty(2 * self.n_iface, dtype=torch.float64, device=dev))
        pack, slot, loc, buf = self._t
        st = torch.cuda.current_stream(dev).cuda_stream
        l = _lib.lib()
        _lib.check(l.fep_gather_f64(dev.index, st, buf.numel(), F_local_t.data_ptr(), pack.data_ptr(), buf.data_ptr()),
                   'fep_gather_f64')
        if self.world > 1:
            if dist.get_backend(group) == 'gloo':                  # rehearsal backend: stage through the host
                tmp = buf.cpu()
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
                buf.copy_(tmp)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        _lib.check(l.fep_scatter_f64(dev.index, st, slot.numel(), buf.data_ptr(), slot.data_ptr(), loc.data_ptr(),
                                     F_local_t.data_ptr()), 'fep_scatter_f64')
        return F_local_t


class ShardedContext(Partition):
    """Partition + the rank's device-resident MeshContext on its local elements."""

    def __init__(self, elements, coordinates, rank, world, dhatp1=None, dhatp2=None, wf=None,
                 element_type=None, device=None):
        super().__init__(elements, coordinates.shape[1], rank, world)
        self.ctx = MeshContext(self.local_elements, np.ascontiguousarray(coordinates[:, self.nodes]), dhatp1, dhatp2, wf,
                               element_type=element_type, device=device)

    def close(self):
        self.ctx.close()

    def set_materials(self, shear, bulk, eta, c):
        """Per-point arrays of the GLOBAL mesh (or scalars); the local slice is uploaded."""
        n_q = self.ctx.n_q

        def loc(v):
            v = np.asarray(v, dtype=np.float64)
            return v if v.ndim == 0 else v.ravel()[self.lo * n_q:self.hi * n_q]
        self.ctx.set_materials(loc(shear), loc(bulk), loc(eta), loc(c))

    def local_point_slice(self):
        return slice(self.lo * self.ctx.n_q, self.hi * self.ctx.n_q)

