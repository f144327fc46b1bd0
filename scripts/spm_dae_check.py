"""Debug/inspection: the singular-mass battery DAE on the lane-per-member banded BDF vs independent oracle solves (tests/test_gpu_diffsl.py has the test)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H
from diffsol_amd import diffsl as fe
import diffsl_models as D
from oracle import oracle as O
O.set_det_pow(True)
tol = dict(rtol=1e-6, atol=[1e-6])
for m_shells, nb, group in ((5, 70, 1), (20, 70, 1), (20, 150, 64)):
    code = D.spm_dae(m_shells)
    p = np.linspace(0.6, 1.4, nb)[:, None]
    t_eval = [600.0, 3000.0, 9000.0, 20000.0]
    m, mid = fe.DiffslModel(code), D.host_model(O, code)
    s = H.Solver(m, p, nbatch=nb, **tol)
    y, tot, mem = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
    yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=group, method=0, **tol)
    ref = O.solve_dense_independent.last_roots
    yo = np.transpose(yo, (1, 0, 2))
    print(m_shells, nb, group, "oracle failed", failed, "status", np.unique(mem["status"], return_counts=True), "stats equal", np.array_equal(mem["stats"].T, so),
          "y equal", np.array_equal(y, yo, equal_nan=True), "roots", np.array_equal(mem["t_root"], ref["t_root"], equal_nan=True), flush=True)
    bad = np.where((mem["stats"].T != so).any(axis=1))[0]
    if len(bad):
        print(" first bad members", bad[:8], "device", mem["stats"].T[bad[0]], "oracle", so[bad[0]])
