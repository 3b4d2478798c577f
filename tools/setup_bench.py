#!/usr/bin/env python3
"""Times the setup stage (row a6): get_elastic_stiffness_matrix on the benchmark mesh, split by phase."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fep = importlib.import_module('fem-elastoplasticity_amd')
hp = importlib.import_module('fem-elastoplasticity_amd.hotpath')
t = sys.argv[1] if len(sys.argv) > 1 else 'P1'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 708
m = fep.square_mesh(N, t, 10)
elem, coord = m['elements'], m['coordinates']
d1, d2, wf = fep.element_tables(t)
n_int = elem.shape[1] * wf.size
G = 1e7 / (2 * 1.48) * np.ones(n_int); Kb = 1e7 / (3 * 0.04) * np.ones(n_int)
fep.MeshContext(fep.square_mesh(4, t, 10)['elements'], fep.square_mesh(4, t, 10)['coordinates']).close()   # warm-up (HIP init)
t0 = time.perf_counter(); ctx = fep.MeshContext(elem, coord, d1, d2, wf); t1 = time.perf_counter()
ctx.set_materials(G, Kb, 1.0, 1.0); K = ctx.step(np.zeros(ctx.n_dof), want=('K',))['K']; t2 = time.perf_counter()
B = hp._strain_displacement_csr(ctx); t3 = time.perf_counter()
D = hp._elastic_D_csr(ctx.geometry()[2], G, Kb); t4 = time.perf_counter()
ctx.close()
t5 = time.perf_counter(); out = fep.get_elastic_stiffness_matrix(elem, coord, G, Kb, d1, d2, wf); t6 = time.perf_counter()
print(f'{t} N={N} n_e={elem.shape[1]}: context (symbolic + geometry kernel) {t1-t0:.2f}s, K_elast (GPU step + download) {t2-t1:.2f}s, '
      f'B (host CSR packaging) {t3-t2:.2f}s, D {t4-t3:.2f}s | get_elastic_stiffness_matrix total {t6-t5:.2f}s '
      f'(reference: 11.5 s for P1 N=708, SURVEY 6)')
