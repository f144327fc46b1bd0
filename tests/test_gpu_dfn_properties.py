"""Checks of front-end-generated models against numbers the front end did NOT produce.

* The single-particle battery model exists twice: as a hand-written registry model (identity mass, n = 42, csrc/dsh_models_dyn.hpp) and as DiffSL text with the terminal
  voltage as an algebraic state (n = 43, tests/diffsl_models.py) that goes through parser, differentiation and HIP code generation.  Same physics, two independent
  implementations: the trajectories must agree to the integration tolerance.
* The reference's 962-state Doyle-Fuller-Newman model (crates/diffsol/benches/pybamm_dfn.diffsl; the text is not part of this repository — the test runs where a copy
  is present: DSH_DFN_MODEL=<path>, _dfn_tmp/pybamm_dfn.diffsl, or the reference checkout): properties that follow from the model's equations, not from our evaluation
  of them — F_0 = F_1 = 0.00018906 (discharge and throughput capacity are exactly linear in t), the terminal voltage stays inside the model's own stop window
  (3.105, 4.1) V and falls monotonically under the constant-current discharge, and every member of a uniform ensemble returns the same bits.
"""
import os

import numpy as np
import pytest

import diffsl_models as D

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


def test_single_particle_model_registry_ode_form_and_diffsl_dae_form_agree(H):
    from diffsol_amd import diffsl as fe
    nb = 96
    cur = np.random.default_rng(12345).uniform(0.6, 1.4, nb)[:, None]
    t_eval = np.linspace(360.0, 3600.0, 10)
    tol = dict(rtol=1e-8, atol=[1e-8])
    ode = H.Solver("spm", cur, nbatch=nb, model_size=20, **tol)
    y_ode, tot_o, mem_o = ode.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    dae = H.Solver(fe.DiffslModel(D.spm_dae(20)), cur, nbatch=nb, **tol)
    y_dae, tot_d, mem_d = dae.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert ode.n == 42 and dae.n == 43 and tot_o["failed_members"] == 0 and tot_d["failed_members"] == 0
    assert np.array_equal(mem_o["root_idx"], mem_d["root_idx"])  # the same members hit the same cut-off
    run = (mem_o["ncols"] == len(t_eval)) & (mem_d["ncols"] == len(t_eval))  # members that reach every save point in both forms
    assert run.sum() >= 8  # most of this sweep reaches a voltage cut-off before 3600 s; their event times are compared below
    yo, yd = np.asarray(y_ode)[:, run, :], np.asarray(y_dae)[:, run, :]  # [nt, members, n]
    m = 20
    # DAE state order: q, thr, negative shells centre -> surface, V, positive shells SURFACE -> CENTRE; registry order: q, thr, negative shells, positive shells centre -> surface
    yd42 = np.concatenate([yd[:, :, :2 + m], yd[:, :, 2 + m + 1:][:, :, ::-1]], axis=2)
    scale = np.abs(yo).max(axis=(0, 1), keepdims=True) + 1e-12
    assert np.max(np.abs(yo - yd42) / scale) < 2e-6
    # and a number neither implementation's integrator can bend: F_0 = I / 3600, so the charge counter is I t / 3600 (spm.ds)
    free = (mem_o["root_idx"] < 0)[run]  # members no cut-off stopped (a stopped member's last column is its state at the event)
    assert free.sum() >= 4
    q_exact = 0.0002777777777777778 * cur[run, 0][None, free] * t_eval[:, None]
    assert np.max(np.abs(yo[:, free, 0] - q_exact)) < 1e-6 and np.max(np.abs(yd[:, free, 0] - q_exact)) < 1e-6
    V = yd[:, :, 2 + m]
    assert np.all(V > 3.105 - 1e-9) and np.all(V < 4.1 + 1e-9) and np.all(np.diff(V[:, free], axis=0) < 0.0)  # discharge: the terminal voltage falls, inside the stop window
    # event times agree to the tolerance as well
    hit = mem_o["root_idx"] >= 0
    if hit.any():
        assert np.max(np.abs(mem_o["t_root"][hit] - mem_d["t_root"][hit]) / mem_o["t_root"][hit]) < 1e-5


def _dfn_path():
    for p in (os.environ.get("DSH_DFN_MODEL"), os.path.join(ROOT, "_dfn_tmp", "pybamm_dfn.diffsl"), "/root/reference/crates/diffsol/benches/pybamm_dfn.diffsl"):
        if p and os.path.exists(p):
            return p
    return None


@pytest.mark.skipif(_dfn_path() is None, reason="the reference's pybamm_dfn.diffsl is not part of this repository (DSH_DFN_MODEL=<path> or _dfn_tmp/pybamm_dfn.diffsl)")
def test_dfn_962_states_capacity_is_linear_voltage_stays_in_its_window_and_falls(H):
    from diffsol_amd import diffsl as fe
    code = open(_dfn_path()).read()
    m = fe.DiffslModel(code)
    assert m.n == 962 and m.has_mass and m.nroots == 2 and m.nout == 1
    nb = 4
    p = np.zeros((nb, 1))
    s = H.Solver(m, p, nbatch=nb, method=H.METHOD_BDF, options=dict(ic_armijo_constant=0.1), rtol=1e-6, atol=[1e-6])
    t_eval = np.linspace(0.0, 3600.0, 100)  # benches/pybamm_dfn.rs
    y, reason = s.solve_dense(list(t_eval))  # [nt, nbatch, n]
    y = np.asarray(y)
    ncols = y.shape[0]
    assert ncols >= 50 and np.isfinite(y).all()
    t = t_eval[:ncols]
    # F_0 = F_1 = 0.00018906: states 0 and 1 (discharge / throughput capacity, A h) are 0.00018906 t whatever the rest of the model does
    for k in (0, 1):
        assert np.max(np.abs(y[:, :, k] - 0.00018906 * t[:, None])) < 1e-6 * (1.0 + 0.00018906 * t[-1])
    # every member of the uniform ensemble: the same bits (lock-step over identical systems)
    assert all(np.array_equal(y[:, 0, :], y[:, b, :]) for b in range(1, nb))
    # out_i = 3.85182... + positive electrode potential at the current collector (the last entry of its 20 cells; constant15_ij picks it): inside the stop window and falling
    pos_pot = y[:, 0, 882:902]
    v = 3.8518235799803934860108256543753668665886 + pos_pot
    v_out = v[:, -1]
    alt = v[:, 0]
    term = v_out if np.all(np.diff(v_out) <= 1e-9) else alt
    assert np.all(term > 3.105 - 1e-6) and np.all(term < 4.1 + 1e-6)
    assert np.all(np.diff(term) <= 1e-9) and term[0] - term[-1] > 0.05
    # concentrations stay physical: particle concentrations positive and below the models' maxima (24983.26 / 51217.93 mol m-3, the clamps in F_i)
    cn, cp_ = y[:, 0, 2:2 + (862 - 2 - 60) // 2], y[:, 0, 2 + (862 - 2 - 60) // 2:862 - 60]
    assert cn.min() > 0.0 and cn.max() < 24983.2619938437 and cp_.min() > 0.0 and cp_.max() < 51217.9257309275
    assert reason in (0, 1, 2)
