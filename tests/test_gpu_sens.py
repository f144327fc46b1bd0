"""Forward sensitivities on the HIP backend: problem.bdf_sens() — Bdf::sensitivity_solve (bdf.rs:934-989), SensEquations
(ode_equations/sens_equations.rs), the sensitivity difference arrays in _update_step_size / _update_diff, sensitivities in the error control — through the
C ABI (dshs_create_sens, dshs_interpolate_sens; device operators dsh_model_rhs_sens / dsh_model_init_sens).  Everything bit for bit against the CPU oracle,
which reproduces all 13 counters of the reference's bdf_test_nalgebra_exponential_decay_sens snapshot (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from helpers import ORACLE_MODEL, robertson_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


def _points(s, pts):
    ys, ss = [], []
    for t in pts:
        while abs(s.scalars()[0] if hasattr(s, "scalars") else s.state()["t"]) < abs(t):
            s.step()
        ys.append(s.interpolate(t))
        ss.append(s.interpolate_sens(t))
    return np.array(ys), np.array(ss)


def test_device_sensitivity_operators_match_the_oracle_models(H, O):
    """dsh_model_rhs_sens / dsh_model_init_sens: df/dp and dy0/dp as n x np matrices in one launch = the oracle's sens_mul / init_sens_mul column by column."""
    from diffsol_amd import _ffi
    import ctypes as C
    L = _ffi.load_device_lib()
    assert L.dsh_model_has_sens(H.MODELS["robertson_ode"], 1) == 1 and L.dsh_model_has_sens(H.MODELS["exponential_decay"], 0) == 1
    assert L.dsh_model_has_sens(H.MODELS["robertson"], 0) == 0 and L.dsh_model_has_sens(H.MODELS["heat1d"], 16) == 0
    nb = 37
    c = H.HipContext(nbatch=nb)
    rng = np.random.default_rng(2)
    for name, size, n, npar in (("robertson_ode", 1, 3, 3), ("exponential_decay", 0, 2, 2)):
        x, p = rng.uniform(0.1, 2.0, (nb, n)), rng.uniform(0.5, 3.0, (nb, npar))
        X, P = H.HipVec.from_vec(x, c), H.HipVec.from_vec(p, c)
        S = H.HipMat.zeros(n, npar, c)
        assert L.dsh_model_rhs_sens(c._h, H.MODELS[name], size, nb, 0.3, X.ptr, P.ptr, S.ptr) == 0
        got = S.to_array()  # [nb, n, np]
        if name == "robertson_ode":
            ref = np.stack([np.stack([-x[:, 0], x[:, 0], 0 * x[:, 0]], 1), np.stack([x[:, 1] * x[:, 2], -(x[:, 1] * x[:, 2]), 0 * x[:, 0]], 1),
                            np.stack([0 * x[:, 0], -(x[:, 1] * x[:, 1]), x[:, 1] * x[:, 1]], 1)], 2)
        else:
            ref = np.stack([np.stack([-x[:, 0], -x[:, 1]], 1), np.zeros((nb, 2))], 2)
        assert np.array_equal(got, ref), name
        assert L.dsh_model_init_sens(c._h, H.MODELS[name], size, nb, 0.0, P.ptr, S.ptr) == 0
        ref0 = np.zeros((nb, n, npar))
        if name == "exponential_decay":
            ref0[:, :, 1] = 1.0
        assert np.array_equal(S.to_array(), ref0)
    S = H.HipMat.zeros(3, 3, c)
    assert L.dsh_model_rhs_sens(c._h, H.MODELS["robertson"], 0, nb, 0.0, X.ptr, P.ptr, S.ptr) < 0  # loud: the DAE model has no parameter derivatives


@pytest.mark.parametrize("error_control", [True, False])
def test_exponential_decay_sensitivities_equal_the_oracle_bitwise_and_reproduce_the_reference_counters(H, O, error_control):
    """The reference's own sensitivity test problem (exponential_decay_problem_sens): single IVP first — the HIP path gives the 13 counters of the reference's
    insta snapshot (bdf.rs:1815-1834) — then a batched ensemble in lock-step, states / sensitivities / counters bit for bit against the oracle."""
    pts = [float(i) for i in range(10)]
    skw = dict(sens=True, sens_rtol=1e-6, sens_atol=[1e-6, 1e-6]) if error_control else dict(sens=True)
    s = H.Solver("exponential_decay", [[0.1, 1.0]], nbatch=1, rtol=1e-6, atol=[1e-6], **skw)
    o = O.OracleSolver(ORACLE_MODEL["exponential_decay"], [0.1, 1.0], rtol=1e-6, atol=[1e-6], **skw)
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and s.stats() == o.stats()
    if error_control:
        st = s.stats()
        assert [st[k] for k in st] == [14, 56, 1, 175, 0, 1, 0, 0, 1, 12, 60, 123, 2]
    t = np.array(pts)[:, None]
    tol = 2e-5 if error_control else 5e-3  # outside the error control nothing bounds the effect of the reference's c = 0 start on the sensitivities
    assert np.abs(ss[:, 1, 0, :] - np.exp(-0.1 * t)).max() < tol and np.abs(ss[:, 0, 0, :] + t * np.exp(-0.1 * t)).max() < tol
    nb = 5
    p = np.stack([0.1 * (np.arange(nb) + 1), 1.0 + np.arange(nb)], axis=1)
    sb = H.Solver("exponential_decay", p, nbatch=nb, rtol=1e-6, atol=[1e-6], **skw)
    ob = O.OracleSolver(ORACLE_MODEL["exponential_decay"], p, nbatch=nb, rtol=1e-6, atol=[1e-6], **skw)
    yb, sbv = _points(sb, pts[:6])
    yob, sob = _points(ob, pts[:6])
    assert np.array_equal(yb, yob) and np.array_equal(sbv, sob) and sb.stats() == ob.stats()
    assert np.array_equal(sb.interpolate_sens(), ob.interpolate_sens())  # state.s


def test_robertson_ensemble_sensitivities_equal_the_oracle_bitwise_and_finite_differences(H, O):
    """Config-2-shaped ensemble (Robertson ODE sweep) with forward sensitivities with respect to (k1, k2, k3), in the error control: lock-step ensemble on the GPU
    = oracle bit for bit (states, 3 x n sensitivities per member, all counters); the sensitivities agree with central differences of plain solves."""
    nb = 6
    p = robertson_params(nb, seed=77)
    kw = dict(model_size=1, rtol=1e-5, atol=[1e-9, 1e-12, 1e-9])
    skw = dict(sens=True, sens_rtol=1e-5, sens_atol=[1e-8])
    pts = [0.4, 4.0, 40.0]
    s = H.Solver("robertson_ode", p, nbatch=nb, **kw, **skw)
    assert not s.fused  # the sensitivity path runs on the trait operations
    o = O.OracleSolver(ORACLE_MODEL["robertson_ode"], p, nbatch=nb, **kw, **skw)
    ys, ss = _points(s, pts)
    yo, so = _points(o, pts)
    assert np.array_equal(ys, yo) and np.array_equal(ss, so) and s.stats() == o.stats()
    tight = dict(model_size=1, rtol=1e-10, atol=[1e-14, 1e-16, 1e-14])
    st = H.Solver("robertson_ode", p, nbatch=nb, sens=True, sens_rtol=1e-10, sens_atol=[1e-12], **tight)
    _, s_t = _points(st, pts)
    for j in range(3):
        dp = 1e-5 * p[:, j]
        pp, pm = p.copy(), p.copy()
        pp[:, j] += dp
        pm[:, j] -= dp
        yp, _ = H.Solver("robertson_ode", pp, nbatch=nb, **tight).solve_to_points(pts)
        ym, _ = H.Solver("robertson_ode", pm, nbatch=nb, **tight).solve_to_points(pts)
        fd = (yp - ym) / (2 * dp)[None, :, None]
        assert np.abs(s_t[:, j] - fd).max() <= 5e-4 * np.abs(fd).max(), j


def test_sensitivity_requests_that_the_backend_cannot_serve_fail_loudly(H):
    with pytest.raises(H.DiffsolHipError):
        H.Solver("robertson", robertson_params(2), nbatch=2, rtol=1e-4, atol=[1e-8, 1e-6, 1e-6], sens=True)            # DAE model: no parameter derivatives
    with pytest.raises(H.DiffsolHipError):
        H.Solver("robertson_ode", robertson_params(2), nbatch=2, model_size=1, method=H.METHOD_TR_BDF2, sens=True)     # BDF only
    s = H.Solver("robertson_ode", robertson_params(2), nbatch=2, model_size=1)
    with pytest.raises(H.DiffsolHipError):
        s.interpolate_sens(0.0)                                                                                         # not created with sensitivities
