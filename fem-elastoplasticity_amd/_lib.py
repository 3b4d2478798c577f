"""
ctypes binding of libfep_hip.so (C ABI: include/fep.h).  There is NO fallback: if the HIP
library is missing, or a call fails, an exception is raised.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

c_double_p = C.POINTER(C.c_double)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_u8_p = C.POINTER(C.c_uint8)
c_void_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); the complete export list of include/fep.h
PROTOTYPES = {
    'fep_version': (C.c_int, []),
    'fep_build_is_ablation': (C.c_int, []),
    'fep_strerror': (C.c_char_p, [C.c_int]),
    'fep_last_hip_error': (C.c_int, []),
    'fep_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'fep_element_shape': (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'fep_malloc': (C.c_int, [C.c_int, c_void_pp, C.c_int64]),
    'fep_free': (C.c_int, [C.c_int, C.c_void_p]),
    'fep_memcpy_h2d': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    'fep_memcpy_d2h': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    'fep_sync': (C.c_int, [C.c_int, C.c_void_p]),
    'fep_host_alloc': (C.c_int, [c_void_pp, C.c_int64]),
    'fep_host_free': (C.c_int, [C.c_void_p]),
    'fep_host_trim': (C.c_int, []),
    'fep_return_map_host': (C.c_int, [C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_return_map_dev': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_create': (C.c_int, [c_void_pp, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_destroy': (C.c_int, [C.c_void_p]),
    'fep_ctx_sizes': (C.c_int, [C.c_void_p, c_i64_p]),
    'fep_ctx_geometry_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_pattern_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_set_materials_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_device_ptr': (C.c_int, [C.c_void_p, C.c_int, c_void_pp]),
    'fep_step_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_step_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_step_host_planar': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_assemble_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_assemble_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_kernel_names': (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]),
    'fep_gather_f64': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_scatter_f64': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_iface_sum_f64': (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_transform_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_transform_host': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_solver_create': (C.c_int, [c_void_pp, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_solver_destroy': (C.c_int, [C.c_void_p]),
    'fep_solver_sizes': (C.c_int, [C.c_void_p, c_i64_p]),
    'fep_solver_spmv_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    'fep_solver_pcg_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                     C.c_int, C.POINTER(C.c_int), c_double_p, C.POINTER(C.c_int)]),
    'fep_solver_amg_clear': (C.c_int, [C.c_void_p]),
    'fep_solver_amg_enable_refresh': (C.c_int, [C.c_void_p]),
    'fep_solver_amg_refresh_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_solver_amg_push_level': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 12 + [C.c_double, C.c_int]),
    'fep_solver_amg_pcg_dev': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                         C.c_int, C.POINTER(C.c_int), c_double_p, C.POINTER(C.c_int)]),
    'fep_aggregate_host': (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, c_i64_p]),
    'fep_spgemm_count_host': (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_spgemm_fill_host': (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'fep_ctx_profile_begin': (C.c_int, [C.c_void_p]),
    'fep_ctx_profile_end': (C.c_int, [C.c_void_p, C.c_void_p, c_double_p, C.POINTER(C.c_int)]),
}


class FepError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        l = lib()
        msg = l.fep_strerror(code).decode()
        if code == -4:
            msg += f' (hipError_t {l.fep_last_hip_error()})'
        super().__init__(f'{where}: {msg} [{code}]')


def lib_path():
    # FEP_LIB_PATH: another build of the same C ABI (in-session A/B of kernel versions, tools/r03_matrix.sh); never a fallback
    return os.environ.get('FEP_LIB_PATH') or _build.LIB


def lib():
    """The loaded library.  Raises ImportError when it has not been built
    (`python __graft_entry__.py build` / `fem-elastoplasticity_amd/build.py`)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(f'{path} is missing: the HIP extension has not been built '
                              f'(run `python -c "import __graft_entry__ as g; g.build()"`). '
                              f'There is no CPU fallback.')
        # torch ships its own copy of the HIP runtime (same soname): whichever is loaded first serves the whole
        # process, and torch finds no GPU when it comes second.  The device-resident entry points take torch
        # tensors' memory, so load torch's runtime first.
        import torch  # noqa: F401
        l = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = l
    return _LIB


def check(code, where):
    if code != 0:
        raise FepError(code, where)


def ptr(a):
    """void* of a NumPy array (or None)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _PinnedBlock:
    """Owner of one page-locked block of the library's cache (fep_host_alloc); the block goes back to the cache when
    the last NumPy view of it is gone."""
    __slots__ = ('ptr',)

    def __init__(self, nbytes):
        p = C.c_void_p()
        code = lib().fep_host_alloc(C.byref(p), nbytes)
        if code != 0:
            raise MemoryError(f'fep_host_alloc({nbytes}): {lib().fep_strerror(code).decode()}')
        self.ptr = p.value

    def __del__(self):
        try:
            if self.ptr:
                lib().fep_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=None):
    """np.empty in page-locked memory: the *_host entry points DMA results straight into it (no staging copy on the
    way back).  The array's base chain ends in a ctypes buffer that owns the block, so slices and views keep it alive
    like any NumPy base and the block returns to the cache with the last of them."""
    import numpy as np
    dt = np.dtype(np.float64 if dtype is None else dtype)
    shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(v) for v in shape)
    n = 1
    for v in shape:
        n *= v
    nbytes = max(n * dt.itemsize, 1)
    try:
        blk = _PinnedBlock(nbytes)
    except MemoryError:
        # no page-locked memory left (the library has already given its idle blocks back and retried): an ordinary array
        # does too — the *_host entry points stage pageable outputs through their pinned ring
        return np.empty(shape, dtype=dt)
    buf = (C.c_char * nbytes).from_address(blk.ptr)
    buf._fep_owner = blk                               # ctypes objects take attributes: ties the block to the buffer
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
