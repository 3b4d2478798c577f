"""
Multi-GPU element sharding: one process per GPU, each owns a contiguous range of
elements and runs the hot path on it with no data-path collective; the only exchange is the sum of
the internal-force contributions on nodes shared between ranks (an RCCL all-reduce over the small
interface vector, `torch.distributed`).  K stays sub-assembled: K = sum_r P_r^T K_r P_r, each rank
holding K_r on its local nodes (rows of interface nodes are partial sums) — what a distributed
Krylov solver consumes with the same interface exchange on its products.

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
        for r, (lo, hi) in enumerate(self.ranges):
            nodes_r = np.unique(elements[:, lo:hi])
            touch[nodes_r] += 1
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

    # ---- host-array exchange (NumPy; used by the gloo tests and small drivers) -----------------
    def exchange_force_host(self, F_local, group=None):
        import torch
        import torch.distributed as dist
        buf = torch.zeros(2 * self.n_iface, dtype=torch.float64)
        buf[torch.from_numpy(self.iface_slot_dofs)] = torch.from_numpy(F_local[self.iface_local_dofs])
        if self.world > 1:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        F_local[self.iface_local_dofs] = buf[torch.from_numpy(self.iface_slot_dofs)].numpy()
        return F_local

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

