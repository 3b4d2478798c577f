"""
On-disk mesh / CSV input and output of the tsx-tunnel flavour.

  load_tsx_mesh        tsx-tunnel/pythonFEM.py:1687-1690: `coord.csv` (2 rows: x, y) and `elem.csv` (3 rows of 1-based
                       vertex ids) -> (coordinates (2, n_n) float64, elements (n_p, n_e) int64 0-based), with the P2 / P4
                       midpoints added as the reference's `create_midpoints` does (TSX:1629-1633 dispatches on the type;
                       P1 — for which the reference returns None, SURVEY C11 — gives the vertices as read)
  dump_free_dof_csv    the dumps the reference keeps beside its driver (k_tangent_qq.csv, f0q.csv, fq.csv: tangent and
                       load / residual vectors restricted to the free DOFs, dense, comma separated) from this package's
                       results, for side-by-side comparison
"""
import os

import numpy as np

from .midpoints import create_midpoints
from .tables import LagrangeElementType, _coerce


def load_tsx_mesh(directory='.', element_type='P1', coord_file='coord.csv', elem_file='elem.csv'):
    """(coordinates, elements) of the CSV mesh in `directory`, elements 0-based (TSX:1687-1688), midpoints per TSX:1690."""
    coords = np.genfromtxt(os.path.join(directory, coord_file), delimiter=',', ndmin=2)
    elem = np.genfromtxt(os.path.join(directory, elem_file), delimiter=',', dtype=int, ndmin=2) - 1
    if coords.shape[0] != 2 or elem.shape[0] != 3:
        raise ValueError(f'expected a 2-row coordinate file and a 3-row element file, got {coords.shape} and {elem.shape}')
    if elem.min() < 0 or elem.max() >= coords.shape[1]:
        raise IndexError('element file refers to nodes the coordinate file does not hold (ids are 1-based on disk)')
    t = _coerce(element_type)
    if t in (LagrangeElementType.P2, LagrangeElementType.P4):
        ext = create_midpoints(t, coords, elem)
        return np.asarray(ext['coord_ext'], dtype=np.float64), np.asarray(ext['elem_ext'], dtype=np.int64)
    if t is not LagrangeElementType.P1:
        raise ValueError('the CSV mesh holds triangles: element_type must be P1, P2 or P4')
    return coords.astype(np.float64), elem.astype(np.int64)


def dump_free_dof_csv(directory, Q, K=None, F0=None, F=None):
    """Writes k_tangent_qq.csv (K[Q][:, Q] dense), f0q.csv (F0[Q]) and fq.csv (F[Q]) — the formats of the files the
    reference ships in tsx-tunnel/ — for whichever of K (sparse, DOF order), F0, F ((2, n_n) or DOF order) is given.
    `Q` is the (2, n_n) boolean mask of free DOFs (TSX:1695-1699)."""
    os.makedirs(directory, exist_ok=True)
    qf = np.asarray(Q, dtype=bool).flatten(order='F')
    if K is not None:
        np.savetxt(os.path.join(directory, 'k_tangent_qq.csv'), K.tocsr()[qf][:, qf].toarray(), delimiter=',')
    for name, v in (('f0q.csv', F0), ('fq.csv', F)):
        if v is not None:
            v = np.asarray(v, dtype=float)
            v = v.flatten(order='F') if v.ndim == 2 else v
            np.savetxt(os.path.join(directory, name), v[qf], delimiter=',')
