"""Names of Elasticity2D/pythonFEM.py on the (elastic) assembly path: `get_elastic_stiffness_matrix`
returns (K, weight) and shifts the 1-based `elements` in place (EL:389, EL:477)."""
from .tables import LagrangeElementType, get_local_basis_volume, get_quadrature_volume   # noqa: F401  EL:49-209
from .hotpath import get_elastic_stiffness_matrix_el as get_elastic_stiffness_matrix     # noqa: F401  EL:368-477
