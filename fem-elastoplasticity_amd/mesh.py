"""
Structured square meshes of the strip-footing benchmark: node / element numbering and
Dirichlet masks identical to the reference generator (`get_nodes_1` DP:63-186,
`get_nodes_2` DP:189-343, `assemble_mesh` DP:354-361), written as closed-form index
arithmetic so that million-element meshes build in milliseconds.

Numbering contract:
  P1/Q1: node (i,j) -> i + (N+1) j; cell (i,j) -> i + N j; P1 elements 2*cell = (V1,V2,V4),
         2*cell+1 = (V2,V3,V4); Q1 = (V1,V2,V3,V4).
  P2:    nodes on the (2N+1)^2 grid, id = i + (2N+1) j; elements (V1,V2,V4,V24,V14,V12),
         (V2,V3,V4,V34,V24,V23).
  Q2:    same grid without the cell centres, numbered row by row (DP:204-209);
         element = (V1,V2,V3,V4,V12,V23,V34,V14).
"""
import numpy as np

from .tables import LagrangeElementType, _coerce


def square_mesh(n_seg, element_type, size_xy=10):
    """Mesh of [0,size_xy]^2 with `n_seg` cells per side.

    Returns the reference's mesh dict: 'coordinates' (2,n_n), 'elements' (n_p,n_e) int64
    0-based, 'surface', 'dirichlet_nodes' (2,n_n), 'Q' (2,n_n) bool.
    """
    return rect_mesh(n_seg, n_seg, element_type, size_xy, size_xy)


def rect_mesh(n_x, n_y, element_type, size_x=10, size_y=10):
    """Same numbering on an n_x by n_y cell rectangle [0,size_x]x[0,size_y] (all four structured element types); used for
    the multi-GPU strips of the benchmark.  Boundary masks follow DP:178-184 with the footing on the top edge
    y == size_y and rollers on x == size_x.  The square n_x == n_y, size_x == size_y is the reference's mesh."""
    return _rect(int(n_x), int(n_y), _coerce(element_type), size_x, size_y)


def _rect(Nx, Ny, t, size_x, size_y):
    ci, cj = np.meshgrid(np.arange(Nx), np.arange(Ny), indexing='xy')    # cell (i,j), j outer
    ci = ci.ravel()
    cj = cj.ravel()
    kx, ky = np.arange(Nx), np.arange(Ny)
    if t in (LagrangeElementType.P1, LagrangeElementType.Q1):
        Mx, My = Nx + 1, Ny + 1
        xs, ys = np.linspace(0, size_x, Mx), np.linspace(0, size_y, My)
        coord = np.array([np.tile(xs, My), np.repeat(ys, Mx)])

        def nid(i, j):
            return i + Mx * j
        V1, V2, V3, V4 = nid(ci, cj), nid(ci + 1, cj), nid(ci + 1, cj + 1), nid(ci, cj + 1)
        if t is LagrangeElementType.P1:
            elem = np.array((V1, V2, V4, V2, V3, V4)).reshape((3, 2 * Nx * Ny), order='F')
        else:
            elem = np.array((V1, V2, V3, V4))
        surf = np.concatenate((np.array((nid(kx, 0), nid(kx + 1, 0))), np.array((nid(Nx, ky), nid(Nx, ky + 1))),
                               np.array((nid(kx, Ny), nid(kx + 1, Ny))), np.array((nid(0, ky), nid(0, ky + 1)))), axis=1)
    elif t in (LagrangeElementType.P2, LagrangeElementType.Q2):
        Mx, My = 2 * Nx + 1, 2 * Ny + 1
        xs, ys = np.linspace(0, size_x, Mx), np.linspace(0, size_y, My)
        if t is LagrangeElementType.P2:
            coord = np.array([np.tile(xs, My), np.repeat(ys, Mx)])

            def nid(i, j):
                return i + Mx * j
        else:
            gi, gj = np.meshgrid(np.arange(Mx), np.arange(My), indexing='xy')
            keep = np.logical_not(np.logical_and(gi % 2 == 1, gj % 2 == 1))
            coord = np.array([xs[gi[keep]], ys[gj[keep]]])

            def nid(i, j):
                # full rows (j even) hold Mx nodes, odd rows only the Nx+1 even-i nodes
                before = ((j + 1) // 2) * Mx + (j // 2) * (Nx + 1)
                return before + np.where(j % 2 == 0, i, i // 2)
        i2, j2 = 2 * ci, 2 * cj
        V1, V2, V3, V4 = nid(i2, j2), nid(i2 + 2, j2), nid(i2 + 2, j2 + 2), nid(i2, j2 + 2)
        V12, V14, V23, V34 = nid(i2 + 1, j2), nid(i2, j2 + 1), nid(i2 + 2, j2 + 1), nid(i2 + 1, j2 + 2)
        if t is LagrangeElementType.P2:
            V24 = nid(i2 + 1, j2 + 1)
            elem = np.array((V1, V2, V4, V24, V14, V12, V2, V3, V4, V34, V24, V23)).reshape((6, 2 * Nx * Ny), order='F')
        else:
            elem = np.array((V1, V2, V3, V4, V12, V23, V34, V14))
        ax, ay = 2 * kx, 2 * ky
        zx, zy = np.zeros(Nx, dtype=np.int64), np.zeros(Ny, dtype=np.int64)
        topx, topy = zy + 2 * Nx, zx + 2 * Ny
        surf = np.concatenate((np.array((nid(ax, zx), nid(ax + 2, zx), nid(ax + 1, zx))),
                               np.array((nid(topx, ay), nid(topx, ay + 2), nid(topx, ay + 1))),
                               np.array((nid(ax, topy), nid(ax + 2, topy), nid(ax + 1, topy))),
                               np.array((nid(zy, ay), nid(zy, ay + 2), nid(zy, ay + 1)))), axis=1)
    else:
        raise ValueError(f'no structured generator for {t}')
    footing = np.logical_and(coord[1, :] == size_y, coord[0, :] <= 1.0001)          # DP:178-184
    dirichlet = np.zeros(coord.shape)
    dirichlet[1, footing] = 1
    Q = coord > 0
    Q[1, footing] = 0
    Q[0, coord[0, :] == size_x] = 0
    return {'coordinates': coord, 'elements': elem.astype(np.int64), 'surface': surf,
            'dirichlet_nodes': dirichlet, 'Q': Q}


def assemble_mesh(level, element_type, size_xy):
    """Reference signature (DP:354-361): N_x = size_xy * 2**level cells per side (DP:67)."""
    return square_mesh(size_xy * 2 ** level, element_type, size_xy)


def renumber_for_locality(elements, coordinates):
    """Node and element numbering along a Morton (Z-order) curve — no counterpart in the reference, whose meshes come
    numbered row by row.  The GPU path gathers node and element data of neighbouring nodes together; a mesh numbered
    at random runs ~7x slower, this restores locality.

    Returns (elements2, coordinates2, node_perm, elem_perm) with coordinates2 = coordinates[:, node_perm],
    elements2 = inverse(node_perm)[elements[:, elem_perm]]: new node i is old node node_perm[i], new element j is old
    element elem_perm[j].  Nodal results map back with  old[:, node_perm] = new,  point results (n_q per element) with
    old.reshape(rows, n_e, n_q)[:, elem_perm] = new.reshape(rows, n_e, n_q)."""
    elements = np.asarray(elements)
    coordinates = np.asarray(coordinates, dtype=float)

    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
        for s, m in ((16, 0x0000FFFF0000FFFF), (8, 0x00FF00FF00FF00FF), (4, 0x0F0F0F0F0F0F0F0F),
                     (2, 0x3333333333333333), (1, 0x5555555555555555)):
            v = (v | (v << np.uint64(s))) & np.uint64(m)
        return v

    def morton(xy):
        lo, hi = xy.min(axis=1, keepdims=True), xy.max(axis=1, keepdims=True)
        q = np.floor((xy - lo) / np.where(hi > lo, hi - lo, 1.0) * 65535.0).astype(np.int64)
        return spread(q[0]) | (spread(q[1]) << np.uint64(1))

    node_perm = np.argsort(morton(coordinates), kind='stable')
    inv = np.empty_like(node_perm)
    inv[node_perm] = np.arange(node_perm.size)
    centroids = coordinates[:, elements].mean(axis=1)
    elem_perm = np.argsort(morton(centroids), kind='stable')
    return inv[elements[:, elem_perm]], coordinates[:, node_perm], node_perm, elem_perm
