"""
Sanitizer builds of the host-only native code (SURVEY section 5, row "race detection / sanitizers"): the threaded
symbolic phase, the tile / gather plans of the P1 kernels, the patch plans of the element route (every element type), the
opt-in node plan of P2 / Q2 and the multigrid aggregation (fem-elastoplasticity_amd/csrc/fep_host.h) are compiled into tests/host_san.cpp with
  g++ -fsanitize=address,undefined   and   g++ -fsanitize=thread
and driven on structured, Delaunay (row order and random numbering), tsx-tunnel and orphan-node meshes.  Every plan
is checked against the mesh by fep_host::validate_p1_plan (all indices the kernels will form stay inside their
tables / LDS regions, tiles partition the blocks, every element has one owner).  CPU only.
"""
import importlib
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

SRC = os.path.join(ROOT, 'tests', 'host_san.cpp')


def _build(tmp, kind):
    exe = os.path.join(tmp, f'host_{kind}')
    flags = {'asan': ['-fsanitize=address,undefined', '-fno-sanitize-recover=all'], 'tsan': ['-fsanitize=thread']}[kind]
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-pthread'] + flags + ['-o', exe, SRC]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    return exe


@pytest.fixture(scope='module')
def binaries(tmp_path_factory):
    if shutil.which('g++') is None:
        pytest.skip('g++ not available')
    tmp = str(tmp_path_factory.mktemp('san'))
    return {k: _build(tmp, k) for k in ('asan', 'tsan')}


def _dump(path, elem, n_n):
    elem = np.ascontiguousarray(elem, dtype=np.int32)
    with open(path, 'wb') as f:
        np.array([elem.shape[0], elem.shape[1], n_n], dtype=np.int32).tofile(f)
        elem.tofile(f)


def _meshes(fep):
    from scipy.spatial import Delaunay
    out = {}
    for t, n in (('P1', 70), ('P2', 16), ('Q1', 30), ('Q2', 12)):
        m = fep.square_mesh(n, t, 10)
        out[f'square_{t}'] = (m['elements'], m['coordinates'].shape[1])
    m = fep.rect_mesh(37, 5, 'P1', 10, 2)                                   # odd number of node rows, short rows
    out['strip_P1'] = (m['elements'], m['coordinates'].shape[1])
    rng = np.random.default_rng(3)
    M = 30
    g = np.stack(np.meshgrid(np.arange(M + 1), np.arange(M + 1), indexing='xy')).reshape(2, -1).astype(float)
    inner = (g[0] > 0) & (g[0] < M) & (g[1] > 0) & (g[1] < M)
    g[:, inner] += rng.uniform(-0.35, 0.35, size=(2, int(inner.sum())))
    out['delaunay_rows'] = (Delaunay(g.T).simplices.T, g.shape[1])
    perm = rng.permutation(g.shape[1])
    tri = Delaunay(g[:, perm].T).simplices.T
    out['delaunay_random'] = (tri[:, rng.permutation(tri.shape[1])], g.shape[1])
    t = load_golden('tsx')
    out['tsx_P1'] = (t['elem'], t['coord'].shape[1])
    out['tsx_P2'] = (t['p2_elem'], t['p2_coord'].shape[1])
    out['tsx_P4'] = (t['p4_elem'], t['p4_coord'].shape[1])
    m = fep.square_mesh(12, 'P1', 10)                                       # nodes 50 and the last belong to no element
    out['orphans_P1'] = (np.where(m['elements'] >= 50, m['elements'] + 1, m['elements']), m['coordinates'].shape[1] + 2)
    out['one_element'] = (np.array([[0], [1], [2]]), 3)
    return out


@pytest.mark.parametrize('kind', ['asan', 'tsan'])
def test_host_native_code_under_sanitizers(fep, binaries, tmp_path, kind):
    env = dict(os.environ, FEP_HOST_THREADS='6', ASAN_OPTIONS='detect_leaks=1', TSAN_OPTIONS='halt_on_error=1')
    for name, (elem, n_n) in _meshes(fep).items():
        path = str(tmp_path / f'{name}.bin')
        _dump(path, elem, n_n)
        for segs in ('2',) if kind == 'tsan' else ('2', '3'):
            res = subprocess.run([binaries[kind], path, segs], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                 timeout=600)
            assert res.returncode == 0 and 'result ok' in res.stdout, (name, res.stdout[-3000:])
            assert 'runtime error' not in res.stdout and 'Sanitizer' not in res.stdout, (name, res.stdout[-3000:])
            if elem.shape[0] == 3:
                assert res.stdout.count(': rc 0 check 0 tiles') == 11, res.stdout  # 4 raw tilings + 7 plan variants, all validated
            # the element route's patch plans: 3 patch sizes x 4 groupings, each replayed against the symbolic phase
            assert len([l for l in res.stdout.splitlines() if l.startswith('patch plan') and ' ok 1 check 0 ' in l]) == 12, res.stdout
            # the multigrid refresh's sparse products on fixed patterns, against the triple loop
            assert 'product plans: rc 0 terms' in res.stdout, res.stdout


def test_two_row_tiles_stage_fewer_elements_on_a_row_numbered_mesh(fep, binaries, tmp_path):
    """The multi-segment tiling is only taken when it pays: on the reference's row-numbered square it stages < 1.8
    elements per owned element against > 2.0 for row strips; on the tsx-tunnel mesh the strips stay."""
    m = fep.square_mesh(70, 'P1', 10)
    path = str(tmp_path / 'sq.bin')
    _dump(path, m['elements'], m['coordinates'].shape[1])
    out = subprocess.run([binaries['asan'], path, '2'], stdout=subprocess.PIPE, text=True).stdout
    lines = {l.split('[')[1].split(']')[0]: l for l in out.splitlines() if l.startswith('p1 plan')}
    per = {k: float(v.split('(')[1].split(' per')[0]) for k, v in lines.items()}
    assert ' segs 2 ' in lines['default'] and per['default'] < 1.8 < 2.0 < per['one segment']
    t = load_golden('tsx')
    path = str(tmp_path / 'tsx.bin')
    _dump(path, t['elem'], t['coord'].shape[1])
    out = subprocess.run([binaries['asan'], path, '2'], stdout=subprocess.PIPE, text=True).stdout
    assert [l for l in out.splitlines() if l.startswith('p1 plan [default]')][0].count(' segs 1 ') == 1


def test_random_meshes_through_the_plan_validator(fep, binaries, tmp_path):
    """Fuzz: Delaunay meshes of random sizes, random node / element numberings, random dropped elements (which leaves
    nodes of no element and irregular degrees), every segment count — each plan must validate, under ASan / UBSan."""
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(20261004)
    for trial in range(6):
        n_pts = int(rng.integers(30, 1500))
        pts = rng.random((n_pts, 2)) * [10.0, rng.uniform(1.0, 10.0)]
        tri = Delaunay(pts).simplices.T
        if trial % 2:
            perm = rng.permutation(n_pts)
            tri = perm[tri]
            tri = tri[:, rng.permutation(tri.shape[1])]
        if trial % 3 == 0:
            tri = tri[:, rng.random(tri.shape[1]) > 0.3]
        path = str(tmp_path / f'fuzz{trial}.bin')
        _dump(path, tri, n_pts)
        for segs in ('1', '2', '4'):
            res = subprocess.run([binaries['asan'], path, segs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            assert res.returncode == 0 and 'result ok' in res.stdout, (trial, segs, res.stdout[-2000:])


@pytest.mark.parametrize('kind', ['asan', 'tsan'])
def test_staging_classes_under_sanitizers(tmp_path, kind):
    """fep_staging.h (pinned cache, copy-thread pool, per-device staging engine) against a host stand-in of its HIP calls
    (tests/staging_san.cpp): two host threads through the process-wide pool, two engines, a copy that fails after a
    device -> host chunk was parked (no pending destination may survive the failed call), double release, allocation
    failure with idle blocks to give back."""
    if shutil.which('g++') is None:
        pytest.skip('g++ not available')
    exe = str(tmp_path / f'staging_{kind}')
    flags = {'asan': ['-fsanitize=address,undefined', '-fno-sanitize-recover=all'], 'tsan': ['-fsanitize=thread']}[kind]
    res = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-pthread'] + flags + ['-o', exe, os.path.join(ROOT, 'tests', 'staging_san.cpp')],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout[-3000:]
    env = dict(os.environ, FEP_COPY_THREADS='6', ASAN_OPTIONS='detect_leaks=1', TSAN_OPTIONS='halt_on_error=1')
    res = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0 and 'staging ok' in res.stdout, res.stdout[-3000:]
    assert 'Sanitizer' not in res.stdout and 'runtime error' not in res.stdout, res.stdout[-3000:]


def test_library_rebuilds_when_the_staging_header_changes(fep):
    """build.needs_build() looks at every file the library is compiled from (fep_staging.h was missing)."""
    import re
    b = importlib.import_module('fem-elastoplasticity_amd.build')
    csrc = os.path.join(ROOT, 'fem-elastoplasticity_amd', 'csrc')
    included = set()
    for src in b.SOURCES:
        todo = [src]
        while todo:
            f = todo.pop()
            for inc in re.findall(r'#include "([^"]+)"', open(os.path.join(csrc, f)).read()):
                name = os.path.normpath(os.path.join(os.path.dirname(f), inc))
                if name not in included and os.path.exists(os.path.join(csrc, name)):
                    included.add(name)
                    todo.append(name)
    deps = {os.path.normpath(d) for d in b.DEPS}
    assert included <= deps, sorted(included - deps)
