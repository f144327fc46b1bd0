#!/usr/bin/env python
"""The reference's largest DiffSL model (crates/diffsol/benches/pybamm_dfn.diffsl: Doyle-Fuller-Newman battery model, 962 states, singular mass matrix;
benches/pybamm_dfn.rs: BDF, default tolerances, ic armijo_constant = 0.1, solve_dense over 100 points to 3600 s) through the HIP backend.  The model text is
NOT part of this repository: pass its path.  Two legs, so the CPU leg can run where there is no GPU and its results travel as numbers:

  python scripts/dfn_gpu.py oracle <model.diffsl> <ref.npz> [t_final] [npoints]   host twin (g++ -O0) through the CPU oracle: operators at y0, solve_dense
  python scripts/dfn_gpu.py gpu    <model.diffsl> <ref.npz> [nbatch]              hiprtc-compiled model on the device: same operators and solve, compared bit for bit

The device form of a model this size is OUTLINED (diffsl.hpp emit_switch): one __noinline__ function per component."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

leg, path, ref_path = sys.argv[1], sys.argv[2], sys.argv[3]
code = open(path).read()
OPT = dict(rtol=1e-6, atol=[1e-6])

if leg == "oracle":
    from oracle import oracle as O
    import diffsl_models as D
    O.build()
    t_final = float(sys.argv[4]) if len(sys.argv) > 4 else 3600.0
    npts = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    t0 = time.perf_counter()
    mid = D.host_model(O, code, opt="-O0")
    t_compile = time.perf_counter() - t0
    n = O.model_dims(mid)["n"]
    p = np.zeros((1, 1))
    y0 = O.model_init(mid, p[0])
    rng = np.random.default_rng(962)
    v = rng.standard_normal(n) * np.maximum(np.abs(y0), 1e-3)
    f0 = O.model_rhs(mid, y0, p[0])
    jv = O.model_jac_mul(mid, y0, p[0], v)
    t_eval = np.linspace(0.0, t_final, npts)
    t0 = time.perf_counter()
    y, st, failed = O.solve_dense_independent(mid, p, t_eval, method=O.METHOD_BDF, options=dict(ic_armijo_constant=0.1), **OPT)  # y [1, nt, n]
    t_solve = time.perf_counter() - t0
    ncols = int(O.solve_dense_independent.last_roots["ncols"][0])
    print(json.dumps(dict(n=n, host_compile_s=t_compile, oracle_solve_s=t_solve, failed=int(failed), ncols=ncols, steps=int(st[0, 0]), newton_iterations=int(st[0, 1]),
                          lu_setups=int(st[0, 2]), error_test_failures=int(st[0, 3]))))
    np.savez(ref_path, y0=y0, v=v, f0=f0, jv=jv, t_eval=t_eval, y=np.asarray(y)[0], ncols=ncols, steps=int(st[0, 0]), oracle_solve_s=t_solve)
else:
    import diffsol_amd as H
    from diffsol_amd import _ffi, diffsl as fe
    ref = np.load(ref_path)
    nb = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    t0 = time.perf_counter()
    m = fe.DiffslModel(code)
    m.precompile(0)
    t_compile = time.perf_counter() - t0
    n = m.n
    L = _ffi.load_device_lib()
    c = H.HipContext(nbatch=nb)
    p = np.zeros((nb, 1))
    X, V, P, Y = H.HipVec.from_vec(np.tile(ref["y0"], (nb, 1)), c), H.HipVec.from_vec(np.tile(ref["v"], (nb, 1)), c), H.HipVec.from_vec(p, c), H.HipVec.zeros(n, c)
    assert L.dsh_model_init(c._h, m.model_id, 0, nb, 0.0, P.ptr, Y.ptr) == 0
    ok_init = np.array_equal(np.asarray(Y.clone_as_vec()).reshape(nb, n), np.tile(ref["y0"], (nb, 1)))
    assert L.dsh_model_rhs(c._h, m.model_id, 0, nb, 0.0, X.ptr, P.ptr, Y.ptr) == 0
    ok_rhs = np.array_equal(np.asarray(Y.clone_as_vec()).reshape(nb, n), np.tile(ref["f0"], (nb, 1)))
    assert L.dsh_model_jac_mul(c._h, m.model_id, 0, nb, 0.0, X.ptr, P.ptr, V.ptr, Y.ptr) == 0
    ok_jv = np.array_equal(np.asarray(Y.clone_as_vec()).reshape(nb, n), np.tile(ref["jv"], (nb, 1)))
    J = H.HipMat.zeros(n, n, c)
    c.sync(); t0 = time.perf_counter()
    assert L.dsh_model_jacobian(c._h, m.model_id, 0, nb, 0.0, X.ptr, P.ptr, J.ptr) == 0
    c.sync(); t_jac = time.perf_counter() - t0
    jd = np.asarray(J.to_array())[0]
    ok_jac = np.array_equal(jd @ np.zeros(n), np.zeros(n)) and np.allclose(jd @ ref["v"], ref["jv"], rtol=1e-9, atol=1e-9 * np.abs(ref["jv"]).max())
    print(json.dumps(dict(leg="operators", n=n, nbatch=nb, device_compile_or_cache_s=t_compile, init_bits=bool(ok_init), rhs_bits=bool(ok_rhs), jac_mul_bits=bool(ok_jv),
                          dense_jacobian_ok=bool(ok_jac), dense_jacobian_ms=t_jac * 1e3)), flush=True)
    t_eval = ref["t_eval"]
    s = H.Solver(m, p, nbatch=nb, method=H.METHOD_BDF, options=dict(ic_armijo_constant=0.1), **OPT)
    t0 = time.perf_counter()
    y, reason = s.solve_dense(list(t_eval))  # [nt, nbatch, n]
    t_solve = time.perf_counter() - t0
    y = np.asarray(y)
    yo = np.asarray(ref["y"])[:, None, :]  # [nt, 1, n]
    same = bool(np.array_equal(y, np.broadcast_to(yo, y.shape)))
    rel = float(np.max(np.abs(y - yo) / (np.abs(yo) + 1e-6)))
    st = s.stats()
    print(json.dumps(dict(leg="solve", n=n, nbatch=nb, t_final=float(t_eval[-1]), npoints=len(t_eval), stop_reason=int(reason), solve_s=t_solve, oracle_solve_s=float(ref["oracle_solve_s"]),
                          steps=int(st["number_of_steps"]), oracle_steps=int(ref["steps"]), lu_setups=int(st["number_of_linear_solver_setups"]),
                          same_bits_as_oracle=same, max_rel_diff=rel)), flush=True)
