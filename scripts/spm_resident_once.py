#!/usr/bin/env python
"""One device-resident solve of the single-particle model (for rocprofv3 passes):  python scripts/spm_resident_once.py [nb] [dae]
"dae": the singular-mass formulation from DiffSL (tests/diffsl_models.py spm_dae(20), n = 43) instead of the built-in identity-mass model."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cur = np.random.default_rng(12345).uniform(0.6, 1.4, (nb, 1))
if len(sys.argv) > 2 and sys.argv[2] == "dae":
    sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/tests")
    import diffsl_models as D
    from diffsol_amd import diffsl
    s = H.Solver(diffsl.DiffslModel(D.spm_dae(20)), cur, nbatch=nb, rtol=1e-6, atol=[1e-6])
else:
    s = H.Solver("spm", cur, nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
y, tot = s.solve_dense_adaptive([600.0, 1800.0, 3600.0], want_host=False, group=1)
print(tot)
