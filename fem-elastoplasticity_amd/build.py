"""
Builds csrc/libfep_hip.so (HIP kernels + C ABI, include/fep.h) in-tree with hipcc for gfx950.
hipcc cross-compiles without a GPU.  Used by __graft_entry__.build() and by `python -m`.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libfep_hip.so')
SOURCES = ['fep_api.hip', 'fep_solver.hip']
DEPS = ['fep_api.hip', 'fep_solver.hip', 'fep_common.h', 'fep_host.h', 'fep_kernels.hip.h', 'fep_staging.h', os.path.join('..', '..', 'include', 'fep.h')]


def hipcc_path():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


LIB_ABLATION = os.path.join(CSRC, 'libfep_hip_abl.so')


def build(force=False, verbose=False, ablation=False):
    """Compile the shared library if it is missing or older than its sources.
    ablation=True: the measurement build (-DFEP_ABLATION: variant switches by environment variable, phase clocks) as
    csrc/libfep_hip_abl.so — never loaded unless FEP_LIB_PATH names it (tools/ only)."""
    lib = LIB_ABLATION if ablation else LIB
    if not force and not ablation and not needs_build():
        return lib
    cmd = [hipcc_path(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] + (['-DFEP_ABLATION'] if ablation else []) + \
          ['-o', lib + '.tmp'] + SOURCES
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout)
    os.replace(lib + '.tmp', lib)
    return lib


if __name__ == '__main__':
    import sys
    print(build(force=True, verbose=True, ablation='--ablation' in sys.argv))
