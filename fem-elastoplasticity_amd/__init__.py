"""
MI355X-native implementation of the hot path of MartinBeseda/FEM-ElastoPlasticity:
Drucker-Prager return map + tangent-stiffness / internal-force assembly, as hand-written
HIP kernels behind a C ABI (include/fep.h), with a Python host layer that keeps the
reference's function names and signatures.

The directory name contains a hyphen; import it with
    fep = importlib.import_module('fem-elastoplasticity_amd')

Flavours of the reference's three `pythonFEM.py` copies:
    fep.plasticity2d_dp   Plasticity2D_DP/pythonFEM.py   (P1, P2, Q1, Q2)
    fep.tsx_tunnel        tsx-tunnel/pythonFEM.py        (adds e0 and P4)
    fep.elasticity2d      Elasticity2D/pythonFEM.py      (elastic K only)
"""
from .tables import (ELEMENT_SHAPE, LagrangeElementType, element_tables, get_local_basis_volume,
                     get_quadrature_volume)
from .mesh import assemble_mesh, rect_mesh, renumber_for_locality, square_mesh
from .hotpath import (MeshContext, assemble_tangent, construct_constitutive_problem,
                      construct_constitutive_problem_tsx, default_device, get_elastic_stiffness_matrix,
                      get_elastic_stiffness_matrix_el)
from ._lib import FepError, lib, lib_path
from .build import build
from .sharding import Partition, ShardedContext, element_ranges
from .newton import solve_strip_footing, solve_tsx_tunnel, transform
from .solver import KrylovSolver, build_amg_hierarchy
from .dist_newton import DistributedPCG, solve_strip_footing_sharded
from .midpoints import create_midpoints, create_midpoints_P2, create_midpoints_P4
from .meshio import dump_free_dof_csv, load_tsx_mesh
from . import plasticity2d_dp, tsx_tunnel, elasticity2d

__all__ = ['LagrangeElementType', 'ELEMENT_SHAPE', 'get_quadrature_volume', 'get_local_basis_volume',
           'element_tables', 'assemble_mesh', 'square_mesh', 'rect_mesh', 'renumber_for_locality', 'Partition', 'ShardedContext', 'element_ranges', 'MeshContext', 'construct_constitutive_problem',
           'construct_constitutive_problem_tsx', 'get_elastic_stiffness_matrix', 'get_elastic_stiffness_matrix_el',
           'assemble_tangent', 'default_device', 'FepError', 'lib', 'lib_path', 'build',
           'solve_strip_footing', 'solve_tsx_tunnel', 'transform', 'KrylovSolver', 'DistributedPCG', 'solve_strip_footing_sharded', 'build_amg_hierarchy', 'create_midpoints', 'create_midpoints_P2',
           'create_midpoints_P4', 'load_tsx_mesh', 'dump_free_dof_csv',
           'plasticity2d_dp', 'tsx_tunnel', 'elasticity2d']
