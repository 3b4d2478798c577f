"""bench.py's contract with the driver, at a small size: `python bench.py [--gpus N]` exits 0 and prints ONE JSON line
with the fields the task statement names (+ roofline, cpu_baseline at N = 1); `--gpus 2` invoked as a plain command starts
its own two ranks (here both on cuda:0 over gloo: FEP_BENCH_SINGLE_DEVICE) in weak and strong mode."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline'}


def _run(args, env=None):
    e = dict(os.environ, **(env or {}))
    e.pop('WORLD_SIZE', None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    j = _run(['--cells', '96', '--steps', '3', '--warmup', '1'])
    assert KEYS <= set(j) and 'cpu_baseline' in j and 'kf_only' in j
    assert j['n_gpus'] == 1 and j['steps'] == 3 and j['warmup'] == 1 and j['dtype'] == 'f64' and j['vs_baseline'] is None
    assert j['unit'] == 'updates/s' and j['higher_is_better'] is True and j['data'] == 'synthetic'
    n = 2 * 96 * 96
    assert j['config']['points_per_gpu'] == n and 'BASELINE configs[3]' not in j['config']['workload']   # only at 708 cells
    assert abs(j['value'] - n * 3 / (j['ms_per_step'] * 3e-3)) <= 1e-6 * j['value']
    r = j['roofline']
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert r['traffic'] is None and abs(r['algorithmic_bytes_per_launch'] - 537.0 * n) < 1e-6     # traffic only for the profiled size
    assert j['cpu_baseline']['kind'] == 'port' and j['cpu_baseline']['cores'] == 1 and j['cpu_baseline']['value'] > 0
    assert 0 < j['config']['smooth_points'] and 0 < j['config']['apex_points']
    # the cpu_baseline leg ran the checker on the very mesh and field of the GPU step: the line carries the comparison
    assert j['parity_max_rel'] <= 1e-11 and j['parity_vs_oracle']['ind_p_mismatches'] == 0


def test_single_gpu_line_p2():
    """--elem P2 (BASELINE configs[4]'s element type, element route): 7 points per element, 461.6 B per update."""
    j = _run(['--elem', 'P2', '--cells', '48', '--steps', '3', '--warmup', '1', '--state', 'random'])
    n_e = 2 * 48 * 48
    assert j['config']['element_type'] == 'P2' and j['config']['points_per_gpu'] == 7 * n_e and j['config']['elements_total'] == n_e
    assert abs(j['value'] - 7 * n_e * 3 / (j['ms_per_step'] * 3e-3)) <= 1e-6 * j['value']
    assert abs(j['roofline']['algorithmic_bytes_per_launch'] - (201 + 96 + 8 * 144 / 7) * 7 * n_e) < 1e-3
    assert 'fixup_kernel' in j['roofline']['kernels_ms'] and j['parity_max_rel'] <= 1e-11
    c = j['config']
    assert min(c['smooth_points'], c['apex_points'], c['points_per_gpu'] - c['smooth_points'] - c['apex_points']) > 0.1 * c['points_per_gpu']


@pytest.mark.parametrize('scaling,et,cells', [('weak', 'P1', 96), ('strong', 'P1', 96), ('strong', 'P2', 48)])
def test_two_ranks_started_by_bench_itself(scaling, et, cells):
    """('strong', 'P2'): the documented configs[4] command at a small size — `--gpus 2 --elem P2 --cells 48 --scaling strong`."""
    j = _run(['--gpus', '2', '--backend', 'gloo', '--elem', et, '--cells', str(cells), '--steps', '3', '--warmup', '1',
              '--scaling', scaling, '--min-elements-per-rank', '0'], env={'FEP_BENCH_SINGLE_DEVICE': '1'})
    assert KEYS <= set(j) and j['n_gpus'] == 2 and j['scaling'] == scaling and 'cpu_baseline' not in j
    n_e = 2 * cells * cells
    nq = {'P1': 1, 'P2': 7}[et]
    assert j['config']['elements_total'] == (n_e if scaling == 'strong' else 2 * n_e) and j['config']['element_type'] == et
    assert j['config']['points_total'] == nq * j['config']['elements_total']
    assert len(j['per_rank']) == 2 and j['active_ranks'] == 2
    # BOTH forms of the interface exchange are timed in the same run; the headline names the one it used
    x = j['exchange']
    assert x['headline'] == 'allreduce' and j['exchange_ms'] == x['alone_ms']
    assert x['alone_ms']['allreduce'] > 0 and x['alone_ms']['p2p'] > 0
    assert set(x['step_ms']) == {'allreduce', 'p2p', 'none'} and min(x['step_ms'].values()) > 0
    assert abs(x['step_ms']['allreduce'] - j['ms_per_step']) < 1e-9
    for m in ('allreduce', 'p2p'):
        assert x['exposed_ms'][m] >= 0 and x['hidden_ms'][m] >= 0 and x['bytes_per_rank'][m] > 0
    assert x['bytes_per_rank']['p2p'] <= x['bytes_per_rank']['allreduce']      # one cut: the same nodes, 8 instead of 16 bytes per DOF slot
    assert ('strong' in j) == (scaling == 'weak')
    if scaling == 'weak':
        assert j['strong']['points_total'] == nq * n_e and j['strong']['elements_total'] == n_e
        assert sum(r['points'] for r in j['strong']['per_rank']) == nq * n_e and j['strong']['exchange']['alone_ms']['p2p'] > 0
    assert abs(j['value'] - j['config']['points_total'] * 3 / (j['ms_per_step'] * 3e-3)) <= 1e-6 * j['value']


def test_small_mesh_stays_on_one_rank():
    """north_star: the collective only when the mesh is large enough — with the default gate (100 000 elements per rank) an
    18 432-element square started on two ranks runs on ONE: rank 1 holds no context, no interface, nothing is exchanged, and
    the line says so."""
    j = _run(['--gpus', '2', '--backend', 'gloo', '--cells', '96', '--steps', '3', '--warmup', '1', '--scaling', 'strong'],
             env={'FEP_BENCH_SINGLE_DEVICE': '1'})
    n = 2 * 96 * 96
    assert j['n_gpus'] == 2 and j['active_ranks'] == 1 and j['config']['points_per_gpu'] == n and j['config']['points_total'] == n
    assert '1 active' in j['config']['parallelism']
    assert j['exchange']['bytes_per_rank'] == {'allreduce': 0, 'p2p': 0}
    assert abs(j['value'] - n * 3 / (j['ms_per_step'] * 3e-3)) <= 1e-6 * j['value']
