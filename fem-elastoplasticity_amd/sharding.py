"""
Multi-GPU element sharding: one process per GPU, each owns a contiguous range of
elements and runs the hot path on it with no data-path collective; the only exchange is the sum of
the internal-force contributions on nodes shared between ranks, `torch.distributed` over RCCL.  K stays
sub-assembled: K = sum_r P_r^T K_r P_r, each rank holding K_r on its local nodes (rows of interface
nodes are partial sums) — what a distributed Krylov solver consumes with the same interface exchange
on its products.

Two forms of the exchange (`exchange=` of the constructors, default from FEP_EXCHANGE, else 'allreduce'; every
`exchange_force*` call can name the other one with `mode=`, which is how bench.py times both in one run):

  'allreduce'   pack kernel (other ranks' slots zero) -> all-reduce over the vector of ALL interface DOFs -> unpack kernel
  'p2p'         neighbour-only: one batch of sends / receives between the ranks on the two sides of a cut
                (persistent buffers, pack = fep_gather_f64, sum in ascending rank order = fep_iface_sum_f64)

`min_elements_per_rank` (north_star: a collective "only when the mesh is large enough"; SURVEY 8e: below ~1e5
elements run on one GPU): only as many ranks take elements as leaves each at least that many — the others hold no
context and take no part in the exchange (`Partition.active` / `.active_world` / `.gated`).

The reference has no parallelism of any kind; this module is new.
"""
import os
import warnings

import numpy as np

from .hotpath import MeshContext


def element_ranges(n_e, world):
    """Contiguous, balanced element ranges [lo, hi) per rank."""
    return [(n_e * r // world, n_e * (r + 1) // world) for r in range(world)]


def _dofs(nodes):
    """DOF ids 2*node + comp of an array of node ids, interleaved."""
    return (2 * np.asarray(nodes, dtype=np.int64)[:, None] + np.arange(2)[None, :]).ravel()


class Partition:
    """Host-only part of the sharding: local element range, local node numbering and the
    interface maps of rank `rank` (no GPU needed).

    global node id of local node i: `nodes[i]`.
    `iface_local` : local node ids that other ranks also touch,
    `iface_slot`  : their positions in the global, sorted list of all interface nodes
                    (the layout of the all-reduced vector, 2 DOFs per slot),
    `mult`        : per local node the number of ranks that hold it (weights 1/mult make inner products of
                    interface-consistent vectors global ones: dist_newton.py),
    `neighbours`  : per rank that shares nodes with this one, the LOCAL DOFs of the shared nodes in ascending global node
                    order (the same order on both sides of a cut).

    Cost: linear in the element table — node sets are flag arrays over the node ids (no sort), every other rank's range is
    read once for the per-node holder counts, and a rank's shared-node list is formed only if its node-id interval
    overlaps this rank's (structured meshes: the ranks r-1 and r+1).  BASELINE configs[4] (4 M P2 elements, 8 M nodes)
    with 8 ranks, measured on one core of the build container: 0.18 s per rank, against 0.79 s for the sort-based version
    of round 3 (np.unique of every range on every rank); identical maps (tests/test_sharding_gloo.py)."""

    def __init__(self, elements, n_n, rank, world, min_elements_per_rank=0, exchange=None):
        elements = np.asarray(elements)
        n_e = int(elements.shape[1])
        n_n = int(n_n)
        self.rank, self.world = int(rank), int(world)
        if self.world > 255:
            raise ValueError('at most 255 ranks (per-node holder counts are bytes)')
        mode = exchange if exchange is not None else os.environ.get('FEP_EXCHANGE', 'allreduce')
        if mode not in ('allreduce', 'p2p'):
            raise ValueError(f"exchange must be 'allreduce' or 'p2p', not {mode!r}")
        self.exchange = mode
        # the gate: ranks beyond `active_world` hold nothing
        aw = self.world
        if min_elements_per_rank and min_elements_per_rank > 0:
            aw = max(1, min(self.world, n_e // int(min_elements_per_rank)))
        self.active_world, self.gated = aw, aw < self.world
        self.active = self.rank < aw
        if self.gated and self.rank == 0:
            warnings.warn(f'{n_e} elements over {self.world} ranks is fewer than {min_elements_per_rank} per rank: '
                          f'{aw} rank(s) take the mesh, the others idle', stacklevel=2)
        self.ranges = element_ranges(n_e, aw) + [(n_e, n_e)] * (self.world - aw)
        self.lo, self.hi = self.ranges[self.rank]

        def flags(lo, hi):
            m = np.zeros(n_n, dtype=bool)
            if hi > lo:
                m[elements[:, lo:hi].ravel()] = True
            return m

        mine_mask = flags(self.lo, self.hi)
        mine = np.flatnonzero(mine_mask)                            # sorted global ids of local nodes
        self.nodes = mine
        g2l = np.zeros(n_n, dtype=np.int32)
        g2l[mine] = np.arange(mine.size, dtype=np.int32)
        self.local_elements = g2l[elements[:, self.lo:self.hi]]
        my_lo, my_hi = (int(mine[0]), int(mine[-1])) if mine.size else (0, -1)
        touch = mine_mask.astype(np.uint8)                          # holders of every node, all ranks
        shared_with = {}
        for r, (lo, hi) in enumerate(self.ranges):
            if r == self.rank or hi <= lo:
                continue
            m = flags(lo, hi)
            touch += m
            if mine.size:
                blk = elements[:, lo:hi]
                if int(blk.min()) <= my_hi and int(blk.max()) >= my_lo:      # else no node in common
                    m &= mine_mask
                    sh = np.flatnonzero(m)
                    if sh.size:
                        shared_with[r] = sh
            del m
        iface_global = np.flatnonzero(touch > 1)
        is_iface = touch[mine] > 1
        self.mult = touch[mine].astype(np.float64)                 # number of ranks that hold each local node
        self.n_iface = int(iface_global.size)
        self.iface_local = np.flatnonzero(is_iface)
        self.iface_slot = np.searchsorted(iface_global, mine[is_iface])
        # DOF-level index vectors (DOF = 2*node + comp)
        self.iface_local_dofs = _dofs(self.iface_local)
        self.iface_slot_dofs = _dofs(self.iface_slot)
        self.neighbours = {r: _dofs(g2l[sh]) for r, sh in shared_with.items()}
        # the neighbour-only exchange, as index tables: segments of the send / receive buffers (ascending rank) and, per
        # interface DOF, its contributions in ascending rank order (-1: this rank's own value, else a position in the receive buffer)
        nb = sorted(self.neighbours)
        self.p2p_ranks = nb
        self.p2p_offsets = np.concatenate([[0], np.cumsum([self.neighbours[r].size for r in nb])]).astype(np.int64)
        self.p2p_send_dofs = (np.concatenate([self.neighbours[r] for r in nb]) if nb else np.zeros(0, np.int64)).astype(np.int32)
        n_if = self.iface_local_dofs.size
        ent_i = [np.arange(n_if, dtype=np.int64)]
        ent_r = [np.full(n_if, self.rank, dtype=np.int64)]
        ent_s = [np.full(n_if, -1, dtype=np.int64)]
        for k, r in enumerate(nb):
            d = self.neighbours[r]
            ent_i.append(np.searchsorted(self.iface_local_dofs, d))
            ent_r.append(np.full(d.size, r, dtype=np.int64))
            ent_s.append(self.p2p_offsets[k] + np.arange(d.size, dtype=np.int64))
        ent_i, ent_r, ent_s = np.concatenate(ent_i), np.concatenate(ent_r), np.concatenate(ent_s)
        order = np.lexsort((ent_r, ent_i))
        self.p2p_src = ent_s[order].astype(np.int32)
        self.p2p_ptr = np.concatenate([[0], np.cumsum(np.bincount(ent_i, minlength=n_if))]).astype(np.int32)
        self._t = None
        self._p2p = None

    # ---- host-array exchange (NumPy; used by the gloo tests and small drivers) -----------------
    def exchange_force_host(self, F_local, group=None, mode=None):
        import torch
        import torch.distributed as dist
        if (mode or self.exchange) == 'p2p':
            if self.world > 1 and self.p2p_ranks:
                send = torch.from_numpy(np.ascontiguousarray(F_local[self.p2p_send_dofs]))
                recv = torch.empty_like(send)
                self._sendrecv(send, recv, group)
                self._iface_sum_host(F_local, recv.numpy())
            return F_local
        if self.n_iface == 0:                                        # (a global quantity: every rank skips the collective)
            return F_local
        buf = torch.zeros(2 * self.n_iface, dtype=torch.float64)
        buf[torch.from_numpy(self.iface_slot_dofs)] = torch.from_numpy(F_local[self.iface_local_dofs])
        if self.world > 1:                                           # ranks the gate left idle take part with zeros
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        F_local[self.iface_local_dofs] = buf[torch.from_numpy(self.iface_slot_dofs)].numpy()
        return F_local

    def _iface_sum_host(self, F_local, recv):
        """0 + c_a + c_b + ... in ascending rank order on every interface DOF (what fep_iface_sum_f64 does on the device)."""
        own = F_local[self.iface_local_dofs].copy()
        acc = np.zeros_like(own)
        ptr, src = self.p2p_ptr, self.p2p_src
        cnt = np.diff(ptr)
        for k in range(int(cnt.max()) if cnt.size else 0):          # k-th contribution of every DOF that has one
            sel = np.flatnonzero(cnt > k)
            s = src[ptr[sel] + k]
            acc[sel] += np.where(s < 0, own[sel], recv[np.maximum(s, 0)] if recv.size else 0.0)
        F_local[self.iface_local_dofs] = acc

    def _sendrecv(self, send, recv, group=None):
        """One batch of sends / receives (ncclGroupStart / End under RCCL): segment k of `send` to rank p2p_ranks[k], its
        counterpart into segment k of `recv`.  Peers of a P2POp are GLOBAL ranks: translated when `group` is a sub-group."""
        import torch.distributed as dist
        ops = []
        for k, r in enumerate(self.p2p_ranks):
            a, b = int(self.p2p_offsets[k]), int(self.p2p_offsets[k + 1])
            peer = dist.get_global_rank(group, r) if group is not None else r
            ops.append(dist.P2POp(dist.isend, send[a:b], peer, group))
            ops.append(dist.P2POp(dist.irecv, recv[a:b], peer, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()                                                 # nccl: orders the current stream behind the batch, no host wait

    # ---- device-resident exchange (torch tensors on the rank's GPU; RCCL) --------------------------
    def exchange_force_(self, F_local_t, group=None, mode=None):
        """In-place: interface DOFs of the local force tensor become the sum over all ranks.  Every kernel and the
        collective / the send-receive batch follow torch's CURRENT stream of the tensor's device (`with
        torch.cuda.stream(s):` selects another one).  No allocation after the first call, no host synchronisation with the
        nccl backend (gloo, the rehearsal backend, stages through pinned host buffers)."""
        import torch
        import torch.distributed as dist
        from . import _lib
        dev = F_local_t.device
        gloo = self.world > 1 and dist.is_initialized() and dist.get_backend(group) == 'gloo'
        if (mode or self.exchange) == 'p2p':
            if self.world == 1 or not self.p2p_ranks:
                return F_local_t
            if self._p2p is None or self._p2p['dev'] != dev:
                n = int(self.p2p_send_dofs.size)
                f64 = dict(dtype=torch.float64)
                self._p2p = {
                    'dev': dev, 'send_idx': torch.from_numpy(self.p2p_send_dofs).to(dev),
                    'loc': torch.from_numpy(self.iface_local_dofs.astype(np.int32)).to(dev),
                    'ptr': torch.from_numpy(self.p2p_ptr).to(dev), 'src': torch.from_numpy(self.p2p_src).to(dev),
                    'send': torch.empty(n, device=dev, **f64), 'recv': torch.empty(n, device=dev, **f64),
                    'send_h': torch.empty(n, **f64).pin_memory() if gloo else None,
                    'recv_h': torch.empty(n, **f64).pin_memory() if gloo else None}
            p = self._p2p
            st = torch.cuda.current_stream(dev).cuda_stream
            l = _lib.lib()
            _lib.check(l.fep_gather_f64(dev.index, st, p['send'].numel(), F_local_t.data_ptr(), p['send_idx'].data_ptr(),
                                        p['send'].data_ptr()), 'fep_gather_f64')
            if gloo:
                p['send_h'].copy_(p['send'])                         # (synchronises: rehearsal only)
                self._sendrecv(p['send_h'], p['recv_h'], group)
                p['recv'].copy_(p['recv_h'], non_blocking=True)
            else:
                self._sendrecv(p['send'], p['recv'], group)
            _lib.check(l.fep_iface_sum_f64(dev.index, st, p['loc'].numel(), p['loc'].data_ptr(), p['ptr'].data_ptr(),
                                           p['src'].data_ptr(), p['recv'].data_ptr(), F_local_t.data_ptr()), 'fep_iface_sum_f64')
            return F_local_t
        if self.n_iface == 0:                                      # (a global quantity: every rank skips the collective)
            return F_local_t
        if not self.active:                                        # a rank the gate left idle: zeros into the all-reduce
            if self._t is None:
                self._t = (torch.zeros(2 * self.n_iface, dtype=torch.float64, device='cpu' if gloo else dev),)
            self._t[0].zero_()
            dist.all_reduce(self._t[0], op=dist.ReduceOp.SUM, group=group)
            return F_local_t
        st = torch.cuda.current_stream(dev).cuda_stream
        l = _lib.lib()
        if self._t is None or self._t[0].device != dev:
            pack = np.full(2 * self.n_iface, -1, dtype=np.int32)
            pack[self.iface_slot_dofs] = self.iface_local_dofs
            self._t = (torch.from_numpy(pack).to(dev), torch.from_numpy(self.iface_slot_dofs.astype(np.int32)).to(dev),
                       torch.from_numpy(self.iface_local_dofs.astype(np.int32)).to(dev),
                       torch.empty(2 * self.n_iface, dtype=torch.float64, device=dev),
                       torch.empty(2 * self.n_iface, dtype=torch.float64).pin_memory() if gloo else None)
        pack, slot, loc, buf, buf_h = self._t
        _lib.check(l.fep_gather_f64(dev.index, st, buf.numel(), F_local_t.data_ptr(), pack.data_ptr(), buf.data_ptr()),
                   'fep_gather_f64')
        if self.world > 1:
            if gloo:                                               # rehearsal backend: stage through the host
                buf_h.copy_(buf)
                dist.all_reduce(buf_h, op=dist.ReduceOp.SUM, group=group)
                buf.copy_(buf_h, non_blocking=True)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        _lib.check(l.fep_scatter_f64(dev.index, st, slot.numel(), buf.data_ptr(), slot.data_ptr(), loc.data_ptr(),
                                     F_local_t.data_ptr()), 'fep_scatter_f64')
        return F_local_t


class ShardedContext(Partition):
    """Partition + the rank's device-resident MeshContext on its local elements (`ctx` is None on a rank the
    `min_elements_per_rank` gate left without elements)."""

    def __init__(self, elements, coordinates, rank, world, dhatp1=None, dhatp2=None, wf=None,
                 element_type=None, device=None, min_elements_per_rank=0, exchange=None):
        super().__init__(elements, coordinates.shape[1], rank, world, min_elements_per_rank, exchange)
        self.ctx = None
        if self.active:
            self.ctx = MeshContext(self.local_elements, np.ascontiguousarray(coordinates[:, self.nodes]), dhatp1, dhatp2, wf,
                                   element_type=element_type, device=device)

    def close(self):
        if self.ctx is not None:
            self.ctx.close()

    def set_materials(self, shear, bulk, eta, c):
        """Per-point arrays of the GLOBAL mesh (or scalars); the local slice is uploaded."""
        if self.ctx is None:
            return
        n_q = self.ctx.n_q

        def loc(v):
            v = np.asarray(v, dtype=np.float64)
            return v if v.ndim == 0 else v.ravel()[self.lo * n_q:self.hi * n_q]
        self.ctx.set_materials(loc(shear), loc(bulk), loc(eta), loc(c))

    def local_point_slice(self):
        n_q = self.ctx.n_q if self.ctx is not None else 0
        return slice(self.lo * n_q, self.hi * n_q)
