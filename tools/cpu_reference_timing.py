"""tools/cpu_reference_timing.py -- BASELINE.md section 2.2, row 1 (build container only: needs /root/reference).

Times the REFERENCE's own aggregation (segmentation/core/models/utils.py voxel_to_aux + aux_to_voxel on its
C++ CPU ops compiled where they lie, oracle/build_ref.py) on the S-uniform frames of SURVEY.md section 8d, next to
oracle/'s scalar restatement and its OpenMP twin on the same cores.  Caveats as for the golden fixtures:
hash_query goes through the oracle restatement (sparsehash absent), and for r = 3 the one spdevoxelize call
is the torch restatement of the CUDA kernel (the reference CPU op hard-wires 8 neighbours)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import make_golden as MG            # imports the reference
from oracle import link_oracle as O
from bench import s_uniform, _cpu_info


def best(fn, k=3):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


model, phys = _cpu_info()
print(f"host: {model}, {phys} physical cores, torch threads {torch.get_num_threads()}, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS', 'unset')}")
for name, n, c, s, r in (("cfg1", 10_000, 16, 7, 3), ("cfg2", 100_000, 64, 7, 3), ("r2", 100_000, 64, 3, 2)):
    coords = s_uniform(n, seed=0)
    x = torch.randn(n, 2 * c, generator=torch.Generator().manual_seed(1))
    t_ref = best(lambda: MG.ref_aggregate(x, coords, s, r))
    xn, cn = x.numpy(), coords.numpy()
    t_port = best(lambda: O.aggregate(xn, cn, s, r))
    O.set_omp(True)
    t_omp = best(lambda: O.aggregate(xn, cn, s, r), 1)
    O.set_omp(False)
    print(f"{name}: N={n} W={2 * c} s={s} r={r}   reference {1e3 * t_ref:8.1f} ms = {n / t_ref:9.3e} vox/s   "
          f"oracle scalar {1e3 * t_port:8.1f} ms = {n / t_port:9.3e} vox/s   oracle OpenMP twin {1e3 * t_omp:8.1f} ms = {n / t_omp:9.3e} vox/s")
