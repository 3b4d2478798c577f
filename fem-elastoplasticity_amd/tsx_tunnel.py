"""Names of tsx-tunnel/pythonFEM.py that sit on the hot path, with identical signatures
(the return map takes the initial strain `e0` as second argument, TSX:990-991)."""
from .tables import LagrangeElementType, get_local_basis_volume, get_quadrature_volume   # noqa: F401  TSX:57-274
from .hotpath import assemble_tangent, get_elastic_stiffness_matrix                      # noqa: F401  TSX:432-542
from .hotpath import construct_constitutive_problem_tsx as construct_constitutive_problem  # noqa: F401  TSX:990-1157
from .midpoints import create_midpoints, create_midpoints_P2, create_midpoints_P4                 # noqa: F401  TSX:1354-1633
