"""
Host-side mirror of the reference's interface for the hot path, on top of the
C ABI (include/fep.h -> csrc/libfep_hip.so).  All arithmetic of the path runs in the HIP
kernels; this module only normalises NumPy layouts, keeps the reference's return
conventions (shapes, aliasing quirks) and wraps CSR values into SciPy matrices.

Reference interface mirrored here (DP = Plasticity2D_DP/pythonFEM.py, TSX = tsx-tunnel/pythonFEM.py):
  construct_constitutive_problem   DP:604-757 / TSX:990-1157
  get_elastic_stiffness_matrix     DP:491-601 / TSX:432-542 / EL:368-477
  (inline) tangent + residual      DP:1047-1058 / TSX:1773-1778  ->  assemble_tangent / MeshContext.step
"""
import ctypes as C
import os

import numpy as np
import scipy.sparse as ssp

from . import _lib
from .tables import ELEMENT_SHAPE, LagrangeElementType, _coerce, element_tables

_NP_TO_TYPE = {3: LagrangeElementType.P1, 6: LagrangeElementType.P2, 4: LagrangeElementType.Q1,
               8: LagrangeElementType.Q2, 15: LagrangeElementType.P4}


def default_device():
    for var in ('FEP_DEVICE', 'LOCAL_RANK'):
        if os.environ.get(var, '') != '':
            return int(os.environ[var])
    return 0


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        a = np.ascontiguousarray(a.reshape(shape))
    return a


def _strain_view(e, n_int):
    """(array kept alive, point stride, component stride) of a (3,n_int) strain in either
    memory order: the driver hands over an F-ordered array (DP:1043), tests C-ordered ones."""
    e = np.asarray(e, dtype=np.float64)
    if e.shape != (3, n_int):
        raise ValueError(f'strain must be (3,{n_int}), got {e.shape}')
    if not (e.flags.c_contiguous or e.flags.f_contiguous):
        e = np.ascontiguousarray(e)
    if n_int == 1 or e.flags.f_contiguous and not e.flags.c_contiguous:
        return np.asfortranarray(e), 3, 1
    return e, 1, n_int


# ---------------------------------------------------------------------------------------
# a2  construct_constitutive_problem
# ---------------------------------------------------------------------------------------
def _return_map(e, e0, ep_prev, shear, bulk, eta, c, apply_plastic_strain, tsx, device=None):
    l = _lib.lib()
    dev = default_device() if device is None else device
    shear = _f64(shear).ravel()
    n_int = shear.size
    bulk, eta, c = _f64(bulk).ravel(), _f64(eta).ravel(), _f64(c).ravel()
    ev, ps, cs = _strain_view(e, n_int)
    e0v = None if e0 is None else _f64(e0).ravel()
    if e0v is not None and e0v.size != 4:
        raise ValueError('e0 must hold 4 values (broadcast (4,1) initial strain, TSX:1052)')
    ep_dev = None
    if ep_prev is not None:
        if ep_prev.shape != (4, n_int):
            raise ValueError(f'ep_prev must be (4,{n_int})')
        # the kernel writes the plastic strain only on accepting calls: otherwise the caller's array is read in place
        direct = (isinstance(ep_prev, np.ndarray) and ep_prev.dtype == np.float64 and ep_prev.flags.c_contiguous
                  and (ep_prev.flags.writeable or not apply_plastic_strain))
        ep_dev = ep_prev if direct else _f64(ep_prev).copy()
    s = _lib.pinned_empty((4, n_int))                  # outputs in page-locked memory: DMA-ed straight into them
    ds = _lib.pinned_empty((9, n_int))
    ind = _lib.pinned_empty(n_int, np.uint8)
    counts = np.zeros(2, dtype=np.int64)
    accept = bool(apply_plastic_strain) and ep_prev is not None
    _lib.check(l.fep_return_map_host(dev, n_int, _lib.ptr(ev), ps, cs, _lib.ptr(e0v), _lib.ptr(ep_dev),
                                     _lib.ptr(shear), _lib.ptr(bulk), _lib.ptr(eta), _lib.ptr(c), int(accept),
                                     _lib.ptr(s), _lib.ptr(ds), _lib.ptr(ind), _lib.ptr(counts)),
               'fep_return_map_host')
    n_smooth, n_apex = int(counts[0]), int(counts[1])
    out = _Result({'s': s, 'ds': ds, 'ind_p': ind.view(np.bool_), 'n_smooth': n_smooth, 'n_apex': n_apex})
    early_out = tsx and n_smooth == 0 and n_apex == 0                  # TSX:1103
    # C2: lambda_final is None in DP always, in TSX whenever a point is plastic
    out['lambda_final'] = np.zeros((1, n_int)) if early_out else None
    if apply_plastic_strain and not early_out:
        if ep_prev is None:
            raise TypeError('apply_plastic_strain needs ep_prev (the reference fails at DP:752 too)')
        if ep_dev is not ep_prev:
            ep_prev[...] = ep_dev                                      # C4: caller's array is mutated
        out['ep'] = ep_prev                                            # and returned as 'ep' (DP:751)
    else:
        out.lazy_zeros('ep', (4, n_int))                               # DP:749: zeros nobody reads on a Newton iterate
    return out


class _Result(dict):
    """The reference's result dict.  A non-accepting call returns 'ep' = zeros((4, n_int)) (DP:749) that the Newton
    iterates never look at; the 32 MB array is created when the key is first touched (any read access to the dict
    that could see it materialises it first), so that the per-iterate call does not pay for it."""
    __slots__ = ('_lazy',)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._lazy = {}

    def lazy_zeros(self, key, shape):
        self._lazy[key] = shape

    def _fill(self):
        if self._lazy:
            for k, shape in self._lazy.items():
                super().__setitem__(k, np.zeros(shape))
            self._lazy = {}

    def __missing__(self, key):
        if key in self._lazy:
            self._fill()
            return super().__getitem__(key)
        raise KeyError(key)

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        super().__setitem__(key, value)

    def __contains__(self, key):
        return key in self._lazy or super().__contains__(key)

    def get(self, key, default=None):
        self._fill()
        return super().get(key, default)

    def keys(self):
        self._fill()
        return super().keys()

    def items(self):
        self._fill()
        return super().items()

    def values(self):
        self._fill()
        return super().values()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()

    def __repr__(self):
        self._fill()
        return super().__repr__()

    def copy(self):
        self._fill()
        return dict(self)


def construct_constitutive_problem(e, ep_prev, shear, bulk, eta, c, apply_plastic_strain=False, device=None):
    """Drop-in for Plasticity2D_DP/pythonFEM.py:604-757 (same positional signature).

    Returns the reference's dict {'s','ds','ind_p','lambda_final','ep'} (+ 'n_smooth',
    'n_apex', the counts the reference logs at DP:730).  `e` is not modified; when
    `apply_plastic_strain` is true `ep_prev` is updated in place and returned as 'ep'.
    """
    return _return_map(e, None, ep_prev, shear, bulk, eta, c, apply_plastic_strain, tsx=False, device=device)


def construct_constitutive_problem_tsx(e, e0, ep_prev, shear, bulk, eta, c, apply_plastic_strain=False, device=None):
    """Drop-in for tsx-tunnel/pythonFEM.py:990-1157: adds the initial strain `e0` (4,1) and the
    all-elastic early-out (zeros for 'lambda_final' and 'ep')."""
    return _return_map(e, e0, ep_prev, shear, bulk, eta, c, apply_plastic_strain, tsx=True, device=device)


# ---------------------------------------------------------------------------------------
# mesh context: static operands + fused step
# ---------------------------------------------------------------------------------------
class MeshContext:
    """Owns the device-resident static operands of one mesh (element table, dphi, weights,
    materials, CSR pattern and gather lists) — what `get_elastic_stiffness_matrix` computes once
    in the reference (DP:977) — and runs the fused hot path on them."""

    def __init__(self, elements, coordinates, dhatp1=None, dhatp2=None, wf=None, element_type=None, device=None):
        l = _lib.lib()
        elements = np.asarray(elements)
        n_p, n_e = elements.shape
        t = _coerce(element_type) if element_type is not None else _NP_TO_TYPE[n_p]
        if ELEMENT_SHAPE[t][0] != n_p:
            raise ValueError(f'{t} needs {ELEMENT_SHAPE[t][0]} nodes per element, got {n_p}')
        n_q = ELEMENT_SHAPE[t][1]
        if dhatp1 is None:
            dhatp1, dhatp2, wf = element_tables(t)
        d1 = np.ascontiguousarray(np.broadcast_to(np.asarray(dhatp1, dtype=np.float64), (n_p, n_q)))
        d2 = np.ascontiguousarray(np.broadcast_to(np.asarray(dhatp2, dtype=np.float64), (n_p, n_q)))
        w = _f64(wf).ravel()
        if w.size != n_q:
            raise ValueError(f'{t} needs {n_q} weight factors')
        coordinates = _f64(coordinates)
        n_n = coordinates.shape[1]
        if elements.size and (elements.min() < 0 or elements.max() >= n_n):
            raise IndexError('element node ids must be 0-based and < n_n')
        el32 = np.ascontiguousarray(elements, dtype=np.int32)
        self.device = default_device() if device is None else device
        self.element_type = t
        self._h = C.c_void_p()
        _lib.check(l.fep_ctx_create(C.byref(self._h), self.device, t.value, n_e, n_n, _lib.ptr(el32),
                                    _lib.ptr(coordinates), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(w)), 'fep_ctx_create')
        sz = (C.c_int64 * 8)()
        _lib.check(l.fep_ctx_sizes(self._h, sz), 'fep_ctx_sizes')
        (self.n_e, self.n_n, self.n_p, self.n_q, self.n_int, self.n_dof, self.nnz, self.n_blk) = [int(v) for v in sz]
        self.elements = el32
        self._pattern = None
        self._geom = None

    # -- lifetime
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            _lib.lib().fep_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    # -- static data
    def geometry(self):
        """(dphi_1, dphi_2, weight, det): (n_p,n_int), (n_p,n_int), (1,n_int), (n_int,)  DP:530-546,585"""
        if self._geom is None:
            d1 = np.empty((self.n_p, self.n_int))
            d2 = np.empty((self.n_p, self.n_int))
            w = np.empty((1, self.n_int))
            det = np.empty(self.n_int)
            _lib.check(_lib.lib().fep_ctx_geometry_host(self._h, _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(w), _lib.ptr(det)),
                       'fep_ctx_geometry_host')
            self._geom = (d1, d2, w, det)
        return self._geom

    def pattern(self):
        """CSR pattern of K on DOFs: (indptr int32 (n_dof+1), indices int32 (nnz)), sorted columns."""
        if self._pattern is None:
            ip = np.empty(self.n_dof + 1, dtype=np.int32)
            ix = np.empty(self.nnz, dtype=np.int32)
            _lib.check(_lib.lib().fep_ctx_pattern_host(self._h, _lib.ptr(ip), _lib.ptr(ix)), 'fep_ctx_pattern_host')
            self._pattern = (ip, ix)
        return self._pattern

    def kernel_names(self, which=0):
        """The kernels one step launches, named as rocprofv3 prints them, joined by ' + ' (which = 0: every output,
        1: the K,F-only step of a Newton iterate)."""
        import ctypes
        buf = ctypes.create_string_buffer(512)
        _lib.check(_lib.lib().fep_ctx_kernel_names(self._h, int(which), buf, 512), 'fep_ctx_kernel_names')
        return buf.value.decode()

    def csr(self, data):
        ip, ix = self.pattern()
        return ssp.csr_matrix((data, ix, ip), shape=(self.n_dof, self.n_dof))

    def set_materials(self, shear, bulk, eta, c):
        a = [_f64(np.broadcast_to(np.asarray(v, dtype=np.float64).ravel(), (self.n_int,))) for v in (shear, bulk, eta, c)]
        _lib.check(_lib.lib().fep_ctx_set_materials_host(self._h, *[_lib.ptr(v) for v in a]), 'fep_ctx_set_materials_host')

    def device_ptr(self, which):
        p = C.c_void_p()
        _lib.check(_lib.lib().fep_ctx_device_ptr(self._h, which, C.byref(p)), 'fep_ctx_device_ptr')
        return p.value

    # -- hot path on host arrays
    def step(self, U, ep_prev=None, e0=None, apply_plastic_strain=False, want=('s', 'ds', 'ind_p', 'K', 'F')):
        """One pass strain -> return map -> tangent -> internal force (DP:1043-1058) for the
        displacement `U` ((2,n_n), or flat DOF order).  Returns a dict with the requested keys among
        'E' (3,n_int), 's' (4,n_int), 'ds' (9,n_int), 'ind_p', 'K' (csr), 'F' (n_dof,), plus
        'n_smooth', 'n_apex'.  `ep_prev` (4,n_int) is updated in place on accept."""
        U = np.asarray(U, dtype=np.float64)
        if U.size != self.n_dof:
            raise ValueError(f'U must hold {self.n_dof} values')
        # the reference's (2, n_n) array goes down as it is: the library interleaves it while staging the transfer
        # (a NumPy `reshape(-1, order='F')` of 1 M nodes costs 2.5 ms, as much as the whole transfer)
        planar = U.ndim == 2 and U.shape[0] == 2 and U.flags.c_contiguous
        u = U if planar else np.ascontiguousarray(U.reshape(-1, order='F') if U.ndim == 2 else U)
        n = self.n_int
        e0v = None if e0 is None else _f64(e0).ravel()
        ep = None
        if ep_prev is not None:
            ok = (ep_prev.dtype == np.float64 and ep_prev.flags.c_contiguous and ep_prev.shape == (4, n))
            ep = ep_prev if ok else _f64(ep_prev, (4, n)).copy()
        out = {}
        E = _lib.pinned_empty((3, n)) if 'E' in want else None          # page-locked: results are DMA-ed straight into them
        s = _lib.pinned_empty((4, n)) if 's' in want else None
        ds = _lib.pinned_empty((9, n)) if 'ds' in want else None
        ind = _lib.pinned_empty(n, np.uint8) if 'ind_p' in want else None
        kd = _lib.pinned_empty(self.nnz) if 'K' in want else None
        F = _lib.pinned_empty(self.n_dof) if 'F' in want else None
        counts = np.zeros(2, dtype=np.int64)
        accept = bool(apply_plastic_strain) and ep is not None
        fn = _lib.lib().fep_step_host_planar if planar else _lib.lib().fep_step_host
        _lib.check(fn(self._h, _lib.ptr(u), _lib.ptr(e0v), _lib.ptr(ep), int(accept), _lib.ptr(E), _lib.ptr(s), _lib.ptr(ds),
                      _lib.ptr(ind), _lib.ptr(kd), _lib.ptr(F), _lib.ptr(counts)), 'fep_step_host')
        if accept and ep is not ep_prev:
            ep_prev[...] = ep
        for k, v in (('E', E), ('s', s), ('ds', ds), ('F', F)):
            if v is not None:
                out[k] = v
        if ind is not None:
            out['ind_p'] = ind.view(np.bool_)
        if kd is not None:
            out['K'] = self.csr(kd)
        out['n_smooth'], out['n_apex'] = int(counts[0]), int(counts[1])
        return out

    def assemble(self, ds=None, s=None):
        """K = B^T blockdiag(w*ds) B and F = B^T (w*s[0:3]) from given point data (DP:1047-1058).
        Returns (K csr or None, F or None)."""
        n = self.n_int
        dsv = None if ds is None else _f64(ds, (9, n))
        sv = None if s is None else _f64(np.asarray(s)[0:3], (3, n))
        kd = _lib.pinned_empty(self.nnz) if ds is not None else None
        F = _lib.pinned_empty(self.n_dof) if s is not None else None
        _lib.check(_lib.lib().fep_assemble_host(self._h, _lib.ptr(dsv), _lib.ptr(sv), _lib.ptr(kd), _lib.ptr(F)),
                   'fep_assemble_host')
        return (None if kd is None else self.csr(kd)), F

    # -- hot path on device-resident arrays (raw device pointers as ints; used by bench.py / torch)
    def step_dev(self, stream, u, ep=0, accept=False, e0=None, e_out=0, s=0, ds=0, ind_p=0, k_data=0, f_out=0, counts=0):
        e0v = None if e0 is None else _f64(e0).ravel()
        _lib.check(_lib.lib().fep_step_dev(self._h, stream, u, _lib.ptr(e0v), ep or None, int(bool(accept)), e_out or None,
                                           s or None, ds or None, ind_p or None, k_data or None, f_out or None,
                                           counts or None), 'fep_step_dev')

    def assemble_dev(self, stream, ds=0, s=0, k_data=0, f_out=0):
        _lib.check(_lib.lib().fep_assemble_dev(self._h, stream, ds or None, s or None, k_data or None, f_out or None),
                   'fep_assemble_dev')

    def transform(self, q_int):
        """Nodal values of an integration-point field (DP:760-816) -> (n_n,) ndarray."""
        q = _f64(q_int).ravel()
        if q.size != self.n_int:
            raise ValueError(f'q_int must hold {self.n_int} values')
        out = np.empty(self.n_n)
        _lib.check(_lib.lib().fep_transform_host(self._h, _lib.ptr(q), _lib.ptr(out)), 'fep_transform_host')
        return out

    def transform_dev(self, stream, q_int, q_node):
        _lib.check(_lib.lib().fep_transform_dev(self._h, stream, q_int, q_node), 'fep_transform_dev')

    def profile_begin(self):
        """Start bracketing every kernel of the following step_dev/assemble_dev calls with HIP events."""
        _lib.check(_lib.lib().fep_ctx_profile_begin(self._h), 'fep_ctx_profile_begin')

    def profile_end(self, stream):
        """-> ({'element': ms, 'csr': ms, 'force': ms} average per launch, n_steps)."""
        ms = (C.c_double * 3)()
        n = C.c_int()
        _lib.check(_lib.lib().fep_ctx_profile_end(self._h, stream, ms, C.byref(n)), 'fep_ctx_profile_end')
        return {'element': ms[0], 'csr': ms[1], 'force': ms[2]}, n.value


# ---------------------------------------------------------------------------------------
# a6  get_elastic_stiffness_matrix
# ---------------------------------------------------------------------------------------
def _strain_displacement_csr(ctx):
    """B (3 n_int x 2 n_n) in canonical CSR exactly as SciPy builds it from the reference's
    triplets (DP:549-571): 6 n_p stored entries per point incl. explicit zeros, sorted columns."""
    d1, d2, _, _ = ctx.geometry()
    n_p, n_q, n_e, n_int = ctx.n_p, ctx.n_q, ctx.n_e, ctx.n_int
    order = np.argsort(ctx.elements, axis=0, kind='stable')                 # columns sorted by node id
    nodes = np.take_along_axis(ctx.elements, order, axis=0).astype(np.int32)
    ordk = np.repeat(order, n_q, axis=1)                                    # (n_p, n_int)
    g1 = np.take_along_axis(d1, ordk, axis=0).T                             # (n_int, n_p)
    g2 = np.take_along_axis(d2, ordk, axis=0).T
    data = np.zeros((n_int, 3, n_p, 2))
    data[:, 0, :, 0] = g1
    data[:, 1, :, 1] = g2
    data[:, 2, :, 0] = g2
    data[:, 2, :, 1] = g1
    cols_e = np.empty((n_e, n_p, 2), dtype=np.int32)
    cols_e[:, :, 0] = 2 * nodes.T
    cols_e[:, :, 1] = 2 * nodes.T + 1
    cols = np.broadcast_to(cols_e[:, None, None, :, :], (n_e, n_q, 3, n_p, 2))
    indptr = np.arange(0, 3 * n_int + 1, dtype=np.int64) * (2 * n_p)
    if indptr[-1] < 2 ** 31:
        indptr = indptr.astype(np.int32)
    return ssp.csr_matrix((data.ravel(), np.ascontiguousarray(cols).ravel(), indptr), shape=(3 * n_int, 2 * ctx.n_n))


def _elastic_D_csr(weight, shear, bulk):
    """D (DP:579-592): block-diagonal elastic tensor times weight, all 9 entries per point stored."""
    n_int = weight.size
    iota = np.array([[1], [1], [0]])
    vol = iota * iota.T
    dev = np.diag([1, 1, 0.5]) - vol / 3
    elast = 2 * dev.reshape((-1, 1)) * shear + vol.reshape((-1, 1)) * bulk      # symmetric: row- = column-major
    vd = elast * (np.ones((9, 1)) * weight.reshape(1, -1))
    base = 3 * np.arange(n_int, dtype=np.int64)
    idx = (base[:, None, None] + np.arange(3)[None, None, :]) + np.zeros((1, 3, 1), dtype=np.int64)
    indptr = 3 * np.arange(3 * n_int + 1, dtype=np.int64)
    return ssp.csr_matrix((vd.T.ravel(), idx.ravel(), indptr), shape=(3 * n_int, 3 * n_int))


def get_elastic_stiffness_matrix(elements, coordinates, shear, bulk, dhatp1, dhatp2, wf, device=None):
    """Drop-in for DP:491-601 / TSX:432-542: returns (K, B, weight, id, jd, D).

    Geometry (Jacobians, dphi, weight) and K_elast = B^T D B are computed on the GPU; `B`, `D`,
    `id`, `jd` are index/packaging work done on the host from the GPU's dphi/weight.  `K` is a CSR
    matrix on the full symbolic pattern (the reference's SciPy product drops numerically-zero
    entries).  The mesh context that produced them rides along as `K.fep_ctx` /
    `B.fep_ctx`; pass either to `assemble_tangent`."""
    ctx = MeshContext(elements, coordinates, dhatp1, dhatp2, wf, device=device)
    shear = _f64(shear).ravel()
    bulk = _f64(bulk).ravel()
    ctx.set_materials(shear, bulk, np.ones(ctx.n_int), np.ones(ctx.n_int))
    # elastic tangent everywhere <=> U = 0 (crit1 = -c < 0): K_elast is one pass of the hot path
    K = ctx.step(np.zeros(ctx.n_dof), want=('K',))['K']
    _, _, weight, _ = ctx.geometry()
    B = _strain_displacement_csr(ctx)
    D = _elastic_D_csr(weight, shear, bulk)
    aux = np.arange(3 * ctx.n_int).reshape((3, ctx.n_int), order='F') + 1      # DP:557
    iD = np.tile(aux, (3, 1))                                                  # DP:589
    jD = np.repeat(aux, 3, axis=0)                                             # DP:590
    K.fep_ctx = ctx
    B.fep_ctx = ctx
    return K, B, weight, iD, jD, D


def get_elastic_stiffness_matrix_el(elements, coordinates, shear, bulk, dhatp1, dhatp2, wf, device=None):
    """Elasticity2D flavour (EL:368-477): `elements` is 1-based and is shifted IN PLACE (EL:389, C10);
    returns only (K, weight)."""
    elements -= 1
    K, _, weight, *_ = get_elastic_stiffness_matrix(elements, coordinates, shear, bulk, dhatp1, dhatp2, wf, device)
    return K, weight


def assemble_tangent(handle, ds, s=None):
    """Replaces the inline DP:1047-1050 (+1058): K_tangent (csr) and F from `ds` (9,n_int) and
    `s` (4,n_int).  `handle` is a MeshContext or any object carrying `.fep_ctx` (the K/B returned
    by get_elastic_stiffness_matrix).  Returns (K_tangent, F); F is None when `s` is None."""
    ctx = handle if isinstance(handle, MeshContext) else handle.fep_ctx
    return ctx.assemble(ds, s)
