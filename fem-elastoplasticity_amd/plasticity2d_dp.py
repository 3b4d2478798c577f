"""Names of Plasticity2D_DP/pythonFEM.py that sit on the hot path, with identical signatures.
`import plasticity2d_dp as pythonFEM` is the drop-in for that module's hot-path functions."""
from .tables import LagrangeElementType, get_local_basis_volume, get_quadrature_volume   # noqa: F401  DP:55-60,364-488
from .mesh import assemble_mesh                                                          # noqa: F401  DP:354-361
from .hotpath import (assemble_tangent, construct_constitutive_problem,                  # noqa: F401  DP:604-757
                      get_elastic_stiffness_matrix)                                      # noqa: F401  DP:491-601
