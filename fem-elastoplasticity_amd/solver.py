"""
Linear solve of one Newton iterate on the GPU ("later a GPU Krylov solver"):
    K[Q][:, Q] dU[Q] = -F[Q]          np.linalg.solve on a dense masked block in the reference, DP:1062-1066 / TSX:1781
Preconditioned conjugate gradients (2x2 node-block Jacobi) in HIP, `fep_solver_*` of include/fep.h.  K is used
where the assembly kernels left it: the `data` array on the context's CSR pattern, in device memory.  torch
tensors are only the holders of device memory here.
"""
import ctypes as C

import numpy as np

from . import _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(torch, device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class KrylovSolver:
    """PCG on the free DOFs `free_dof` (bool, DOF order = Q.flatten(order='F')) of the pattern of `ctx`
    (a MeshContext) or of an explicit `(indptr, indices)` pair with `device`."""

    def __init__(self, ctx, free_dof, device=None):
        import torch
        if hasattr(ctx, 'pattern'):
            ip, ix = ctx.pattern()
            device = ctx.device if device is None else device
        else:
            ip, ix = ctx
        ip = np.ascontiguousarray(ip, dtype=np.int32)
        ix = np.ascontiguousarray(ix, dtype=np.int32)
        fd = np.ascontiguousarray(np.asarray(free_dof).ravel() != 0, dtype=np.uint8)
        if fd.size != ip.size - 1 or fd.size % 2:
            raise ValueError('free_dof must hold one flag per DOF (2 per node)')
        self.device = 0 if device is None else device
        self._torch = torch
        self._dev = torch.device('cuda', self.device)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().fep_solver_create(C.byref(self._h), self.device, fd.size // 2, _lib.ptr(ip), _lib.ptr(ix),
                                                _lib.ptr(fd)), 'fep_solver_create')
        sz = (C.c_int64 * 4)()
        _lib.check(_lib.lib().fep_solver_sizes(self._h, sz), 'fep_solver_sizes')
        self.n_n, self.n_dof, self.nnz, self.n_free = [int(v) for v in sz]
        self.free_dof = fd.view(np.bool_)
        self.last = None

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            _lib.lib().fep_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, v, n):
        torch = self._torch
        if not isinstance(v, torch.Tensor):
            v = torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()))
        v = v.to(self._dev)
        if v.dtype != torch.float64 or v.numel() != n or not v.is_contiguous():
            raise ValueError(f'expected {n} contiguous float64 values')
        return v

    def spmv(self, k_data, x, out=None, masked=False):
        """y = K x (masked: rows of constrained DOFs zeroed; x must be 0 there).  Device tensors in and out."""
        torch = self._torch
        k = self._vec(k_data, self.nnz)
        x = self._vec(x, self.n_dof)
        y = torch.empty(self.n_dof, dtype=torch.float64, device=self._dev) if out is None else out
        _lib.check(_lib.lib().fep_solver_spmv_dev(self._h, _stream(torch, self._dev), _p(k), _p(x), _p(y), int(masked)),
                   'fep_solver_spmv_dev')
        return y

    def pcg(self, k_data, b, out=None, rtol=1e-12, max_iter=100000, check_every=50):
        """Solves K[Q][:,Q] x[Q] = b[Q]; returns the full-length device tensor x (0 on constrained DOFs).
        `self.last` = {'iters', 'relres', 'state'} with state 1 = converged, 0 = max_iter, 2 = breakdown."""
        torch = self._torch
        k = self._vec(k_data, self.nnz)
        b = self._vec(b, self.n_dof)
        x = torch.empty(self.n_dof, dtype=torch.float64, device=self._dev) if out is None else out
        it, st, rr = C.c_int(), C.c_int(), C.c_double()
        _lib.check(_lib.lib().fep_solver_pcg_dev(self._h, _stream(torch, self._dev), _p(k), _p(b), _p(x), float(rtol),
                                                 int(max_iter), int(check_every), C.byref(it), C.byref(rr),
                                                 C.byref(st)), 'fep_solver_pcg_dev')
        self.last = {'iters': it.value, 'relres': rr.value, 'state': st.value}
        return x

    def solve_host(self, K, b, **kw):
        """Convenience for host data: `K` a csr_matrix on the pattern (or its data array), `b` (n_dof,) -> ndarray."""
        data = K.data if hasattr(K, 'data') and hasattr(K, 'indptr') else K
        return self.pcg(np.asarray(data, dtype=np.float64), b, **kw).cpu().numpy()
