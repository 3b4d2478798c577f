"""
P1 -> P2 / P4 mesh enrichment with the reference's node numbering.

Mirrors `create_midpoints_P2` (tsx-tunnel/pythonFEM.py:1508-1626), `create_midpoints_P4` (TSX:1354-1505)
and the dispatcher `create_midpoints` (TSX:1629-1633): same names, same returned keys, same ids for the new
nodes (allocated in element-visit order, first element to see an edge creates its nodes), same local node
order — but the neighbour across an edge comes from an edge -> elements map built once (O(n_e)) instead
of two `np.where` scans of the whole element table per edge (O(n_e^2)).

Local orders: P2 rows 3..5 = midpoints of (V2V3, V3V1, V1V2); P4 rows 3..5 = midpoints of
(V1V2, V2V3, V3V1), rows 6..11 = quarter points (two per edge, nearer the edge's first vertex first),
rows 12..14 = interior nodes nearest V1, V2, V3.
"""
import numpy as np

from .tables import LagrangeElementType, _coerce


def _edge_map(elem):
    """undirected edge (min, max) -> list of elements that contain both vertices, in element order."""
    m = {}
    n_e = elem.shape[1]
    for i in range(n_e):
        v = (int(elem[0, i]), int(elem[1, i]), int(elem[2, i]))
        for a, b in ((v[0], v[1]), (v[1], v[2]), (v[2], v[0])):
            m.setdefault((a, b) if a < b else (b, a), []).append(i)
    return m


def _neighbour(m, a, b, i):
    """the element other than `i` that contains the edge {a,b} (None on the boundary)."""
    for j in m[(a, b) if a < b else (b, a)]:
        if j != i:
            return j
    return None


def create_midpoints_P2(coord, elem):
    """TSX:1508-1626.  Returns 'coord_mid', 'surf', 'coord_ext', 'elem_ext', 'elem_ed', 'edge_el'."""
    coord = np.asarray(coord, dtype=float)
    elem = np.asarray(elem)
    n_e, n_n = elem.shape[1], coord.shape[1]
    m = _edge_map(elem)
    coord_mid = np.zeros((2, 3 * n_e))
    elem_mid = np.zeros((3, n_e))
    elem_ed = np.zeros((3, n_e))
    edge_el = np.zeros((2, 3 * n_e))
    surf = np.zeros((3, 3 * n_e))
    ind = 0
    ind_s = 0
    # slot s of an element = its edge (A, B); the neighbour takes the midpoint at the slot given by the
    # position of B among ITS vertices (TSX:1546-1554): first -> slot 2, second -> slot 0, third -> slot 1
    slot_of_pos = (2, 0, 1)
    for i in range(n_e):
        V = (int(elem[0, i]), int(elem[1, i]), int(elem[2, i]))
        for s, (A, B) in enumerate(((V[1], V[2]), (V[2], V[0]), (V[0], V[1]))):    # TSX:1530, 1561, 1591
            if elem_mid[s, i] != 0:
                continue
            coord_mid[:, ind] = (coord[:, A] + coord[:, B]) / 2
            elem_mid[s, i] = n_n + ind
            elem_ed[s, i] = ind
            edge_el[0, ind] = i
            j = _neighbour(m, A, B, i)
            if j is not None:
                edge_el[1, ind] = j
                vj = (int(elem[0, j]), int(elem[1, j]), int(elem[2, j]))
                sj = slot_of_pos[0 if B == vj[0] else (1 if B == vj[1] else 2)]
                elem_mid[sj, j] = n_n + ind
                elem_ed[sj, j] = ind
            else:
                surf[:, ind_s] = (B, A, n_n + ind)
                ind_s += 1
            ind += 1
    coord_mid = coord_mid[:, 0:ind]
    return {'coord_mid': coord_mid, 'surf': surf[:, 0:ind_s], 'coord_ext': np.concatenate([coord, coord_mid], axis=1),
            'elem_ext': np.array(np.concatenate([elem, elem_mid], axis=0), dtype=int),
            'elem_ed': elem_ed, 'edge_el': edge_el[:, 0:ind]}


def create_midpoints_P4(coord, elem):
    """TSX:1354-1505.  Returns 'coord_mid', 'surf', 'coord_ext', 'elem_ext'."""
    coord = np.asarray(coord, dtype=float)
    elem = np.asarray(elem)
    n_e, n_n = elem.shape[1], coord.shape[1]
    m = _edge_map(elem)
    coord_mid = np.zeros((2, 12 * n_e))
    elem_mid = np.zeros((12, n_e))
    surf = np.zeros((5, 3 * n_e))
    ind = -1
    ind_s = -1
    for i in range(n_e):
        V1, V2, V3 = int(elem[0, i]), int(elem[1, i]), int(elem[2, i])
        c1, c2, c3 = coord[:, V1], coord[:, V2], coord[:, V3]
        coord_mid[:, ind + 1] = c1 / 2 + c2 / 4 + c3 / 4                        # TSX:1374-1381
        coord_mid[:, ind + 2] = c1 / 4 + c2 / 2 + c3 / 4
        coord_mid[:, ind + 3] = c1 / 4 + c2 / 4 + c3 / 2
        elem_mid[9, i], elem_mid[10, i], elem_mid[11, i] = n_n + ind + 1, n_n + ind + 2, n_n + ind + 3
        ind += 3
        for s, (A, B) in enumerate(((V1, V2), (V2, V3), (V3, V1))):            # TSX:1386, 1424, 1463
            if elem_mid[s, i] != 0:
                continue
            cA, cB = coord[:, A], coord[:, B]
            coord_mid[:, ind + 1] = (cA + cB) / 2
            coord_mid[:, ind + 2] = 3 * cA / 4 + cB / 4
            coord_mid[:, ind + 3] = cA / 4 + 3 * cB / 4
            elem_mid[s, i] = n_n + ind + 1
            elem_mid[3 + 2 * s, i] = n_n + ind + 2
            elem_mid[4 + 2 * s, i] = n_n + ind + 3
            j = _neighbour(m, A, B, i)
            if j is not None:
                vj = (int(elem[0, j]), int(elem[1, j]), int(elem[2, j]))
                sj = 0 if B == vj[0] else (1 if B == vj[1] else 2)              # TSX:1405-1416
                elem_mid[sj, j] = n_n + ind + 1
                elem_mid[3 + 2 * sj, j] = n_n + ind + 3                         # the neighbour walks the edge backwards
                elem_mid[4 + 2 * sj, j] = n_n + ind + 2
            else:
                ind_s += 1
                surf[:, ind_s] = (B, A, n_n + ind + 1, n_n + ind + 2, n_n + ind + 3)
            ind += 3
    coord_mid = coord_mid[:, 0:ind + 1]
    return {'coord_mid': coord_mid, 'surf': surf[:, 0:ind_s + 1], 'coord_ext': np.concatenate([coord, coord_mid], axis=1),
            'elem_ext': np.array(np.concatenate([elem, elem_mid], axis=0), dtype=int)}


def create_midpoints(elem_type, coord, elem):
    """TSX:1629-1633 (returns None for element types without midpoints, like the reference)."""
    t = _coerce(elem_type)
    if t is LagrangeElementType.P2:
        return create_midpoints_P2(coord, elem)
    if t is LagrangeElementType.P4:
        return create_midpoints_P4(coord, elem)
    return None
