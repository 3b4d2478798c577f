"""
Reference-element tables: quadrature rules and shape-function
derivative tables consumed by the geometry kernel.

Mirrors `get_quadrature_volume` (DP:364-412, TSX:67-128) and
`get_local_basis_volume` (DP:415-488, TSX:155-274): same names, same argument
meaning, same return shapes (xi (2,n_q), wf (1,n_q); hatp/dhatp (n_p,n_q),
P1 derivative tables are (3,1) columns).  Values are checked against tables
captured from the reference in tests/test_host_tables_mesh.py.

Local node orders: P2 = 3 vertices then midpoints of the edges
opposite to vertex 1,2,3; Q2 = 4 vertices then 4 edge midpoints (serendipity);
P4 = 3 vertices, 3 edge midpoints (V1V2, V2V3, V3V1), 6 quarter points,
3 interior nodes (TSX:227-241).
"""
import enum

import numpy as np


class LagrangeElementType(enum.Enum):
    """Same members and values as the reference enum (DP:55-60, TSX:57-63)."""
    P1 = 1
    P2 = 2
    Q1 = 3
    Q2 = 4
    P4 = 5


#: (n_p, n_q) per element type
ELEMENT_SHAPE = {LagrangeElementType.P1: (3, 1), LagrangeElementType.P2: (6, 7),
                 LagrangeElementType.Q1: (4, 4), LagrangeElementType.Q2: (8, 9),
                 LagrangeElementType.P4: (15, 12)}


def _coerce(el_type):
    """Accept our enum, the reference's enum (any Enum with the same .name) or the name."""
    if isinstance(el_type, LagrangeElementType):
        return el_type
    name = getattr(el_type, 'name', el_type)
    if isinstance(name, str) and name in LagrangeElementType.__members__:
        return LagrangeElementType[name]
    return LagrangeElementType(el_type)


def get_quadrature_volume(el_type):
    """Quadrature points `xi` (2,n_q) and weight factors `wf` (1,n_q).  DP:364-412, TSX:67-128.

    Quirks kept: the Q2 rule uses the 3x3 Gauss weights at +-1/sqrt(3) (DP:407-410);
    the P4 rule carries the digit typo 0.06308901449102 (TSX:114).
    """
    t = _coerce(el_type)
    g = 1 / np.sqrt(3)
    if t is LagrangeElementType.P1:
        return np.array([[1 / 3], [1 / 3]]), np.array([[0.5]])
    if t is LagrangeElementType.P2:
        a, b, c, d = 0.1012865073235, 0.7974269853531, 0.4701420641051, 0.0597158717898
        xi = np.array([[a, b, a, c, c, d, 1 / 3],
                       [a, a, b, d, c, c, 1 / 3]])
        w1, w2 = 0.1259391805448, 0.1323941527885
        return xi, 0.5 * np.array([[w1, w1, w1, w2, w2, w2, 0.225]])
    if t is LagrangeElementType.Q1:
        return np.array([[-g, -g, g, g], [-g, g, -g, g]]), np.array([[1, 1, 1, 1]])
    if t is LagrangeElementType.Q2:
        xi = np.array([[-g, g, g, -g, 0, g, 0, -g, 0],
                       [-g, -g, g, g, -g, 0, g, 0, 0]])
        wc, we, wm = 25 / 81, 40 / 81, 64 / 81
        return xi, np.array([[wc, wc, wc, wc, we, we, we, we, wm]])
    if t is LagrangeElementType.P4:
        a, a_typo, b = 0.063089014491502, 0.06308901449102, 0.873821971016996
        c, d = 0.249286745170910, 0.501426509658179
        f, h, m = 0.310352451033785, 0.053145049844816, 0.636502499121399
        xi = np.array([[a, a_typo, b, c, c, d, f, f, h, h, m, m],
                       [a, b, a, c, d, c, h, m, f, m, f, h]])
        w1, w2, w3 = 0.050844906370207, 0.116786275726379, 0.082851075618374
        return xi, np.array([[w1, w1, w1, w2, w2, w2, w3, w3, w3, w3, w3, w3]]) / 2
    raise ValueError(el_type)


# P4 (quartic Lagrange triangle): every basis function is a product of the 1-D factors
# g_m(l) = prod_{k<m} (4 l - k) in the barycentric coordinates, times a constant.
_P4_TERMS = [  # (constant, (m0, m1, m2)) in the reference's node order, TSX:227-241
    (1 / 6, (4, 0, 0)), (1 / 6, (0, 4, 0)), (1 / 6, (0, 0, 4)),
    (4, (2, 2, 0)), (4, (0, 2, 2)), (4, (2, 0, 2)),
    (8 / 3, (3, 1, 0)), (8 / 3, (1, 3, 0)), (8 / 3, (0, 3, 1)),
    (8 / 3, (0, 1, 3)), (8 / 3, (1, 0, 3)), (8 / 3, (3, 0, 1)),
    (32, (2, 1, 1)), (32, (1, 2, 1)), (32, (1, 1, 2)),
]


def _g(lam, m):
    """g_m(lam) = lam (4 lam - 1) ... (4 lam - (m-1)) and its derivative w.r.t. lam."""
    val = np.ones_like(lam)
    der = np.zeros_like(lam)
    for k in range(m):
        fac = lam if k == 0 else 4 * lam - k
        dfac = 1.0 if k == 0 else 4.0
        der = der * fac + val * dfac
        val = val * fac
    return val, der


def _p4_basis(x1, x2):
    lam = (1 - x1 - x2, x1, x2)
    dlam = ((-1.0, -1.0), (1.0, 0.0), (0.0, 1.0))      # d lam_i / d (xi_1, xi_2)
    hat, d1, d2 = [], [], []
    for cst, ms in _P4_TERMS:
        gv = [_g(lam[i], ms[i]) for i in range(3)]
        hat.append(cst * gv[0][0] * gv[1][0] * gv[2][0])
        grads = []
        for j in range(2):
            acc = 0.0
            for i in range(3):
                term = gv[i][1] * dlam[i][j]
                for o in range(3):
                    if o != i:
                        term = term * gv[o][0]
                acc = acc + term
            grads.append(cst * acc)
        d1.append(grads[0])
        d2.append(grads[1])
    return np.array(hat), np.array(d1), np.array(d2)


def get_local_basis_volume(el_type, xi):
    """Basis functions and their xi_1 / xi_2 derivatives at the points `xi` (2,n_q).
    Returns (hatp, dhatp1, dhatp2), each (n_p, n_q) except the P1 derivative
    tables, which are constant (3,1) columns as in the reference (DP:442-443)."""
    t = _coerce(el_type)
    x1 = np.asarray(xi[0], dtype=float)
    x2 = np.asarray(xi[1], dtype=float)
    x0 = 1 - x1 - x2
    zero = np.zeros(np.size(x1, 0))
    if t is LagrangeElementType.P1:
        return np.array([x0, x1, x2]), np.array([[-1], [1], [0]]), np.array([[-1], [0], [1]])
    if t is LagrangeElementType.P2:
        hat = np.array([x0 * (2 * x0 - 1), x1 * (2 * x1 - 1), x2 * (2 * x2 - 1),
                        4 * x1 * x2, 4 * x0 * x2, 4 * x0 * x1])
        d1 = np.array([1 - 4 * x0, 4 * x1 - 1, zero, 4 * x2, -4 * x2, 4 * (x0 - x1)])
        d2 = np.array([1 - 4 * x0, zero, 4 * x2 - 1, 4 * x1, 4 * (x0 - x2), -4 * x1])
        return hat, d1, d2
    if t is LagrangeElementType.Q1:
        m1, p1, m2, p2 = 1 - x1, 1 + x1, 1 - x2, 1 + x2
        hat = np.array([m1 * m2 / 4, p1 * m2 / 4, p1 * p2 / 4, m1 * p2 / 4])
        d1 = np.array([-m2 / 4, m2 / 4, p2 / 4, -p2 / 4])
        d2 = np.array([-m1 / 4, -p1 / 4, p1 / 4, m1 / 4])
        return hat, d1, d2
    if t is LagrangeElementType.Q2:
        m1, p1, m2, p2 = 1 - x1, 1 + x1, 1 - x2, 1 + x2
        q1, q2 = 1 - pow(x1, 2), 1 - pow(x2, 2)
        hat = np.array([m1 * m2 * (-1 - x1 - x2) / 4, p1 * m2 * (-1 + x1 - x2) / 4,
                        p1 * p2 * (-1 + x1 + x2) / 4, m1 * p2 * (-1 - x1 + x2) / 4,
                        q1 * m2 / 2, p1 * q2 / 2, q1 * p2 / 2, m1 * q2 / 2])
        d1 = np.array([m2 * (2 * x1 + x2) / 4, m2 * (2 * x1 - x2) / 4,
                       p2 * (2 * x1 + x2) / 4, p2 * (2 * x1 - x2) / 4,
                       -x1 * m2, q2 / 2, -x1 * p2, -q2 / 2])
        d2 = np.array([m1 * (x1 + 2 * x2) / 4, p1 * (-x1 + 2 * x2) / 4,
                       p1 * (x1 + 2 * x2) / 4, m1 * (-x1 + 2 * x2) / 4,
                       -q1 / 2, -p1 * x2, q1 / 2, -m1 * x2])
        return hat, d1, d2
    if t is LagrangeElementType.P4:
        return _p4_basis(x1, x2)
    raise ValueError(el_type)


def element_tables(el_type):
    """(dhatp1, dhatp2, wf) as C-contiguous (n_p,n_q), (n_p,n_q), (n_q,) float64 arrays —
    the form the C ABI takes (include/fep.h: fep_ctx_create)."""
    t = _coerce(el_type)
    xi, wf = get_quadrature_volume(t)
    _, d1, d2 = get_local_basis_volume(t, xi)
    n_p, n_q = ELEMENT_SHAPE[t]
    d1 = np.ascontiguousarray(np.broadcast_to(np.asarray(d1, dtype=float), (n_p, n_q)))
    d2 = np.ascontiguousarray(np.broadcast_to(np.asarray(d2, dtype=float), (n_p, n_q)))
    return d1, d2, np.ascontiguousarray(np.asarray(wf, dtype=float).ravel())
