"""OdeSolverMethod::solve — the state after EVERY accepted step (crates/diffsol/src/ode_solver/method.rs:227-258 over :881-961) — on the device-resident BDF
(VERDICT r3 missing 5: the resident path only had solve_dense): dsh_bdf_solve_adaptive_steps / dshs_solve_adaptive / Solver.solve_adaptive.  The checker restates the
reference's loop over the oracle's stepping solver, one independent IVP per member (group = 1) or one 64-member batch (group = 64): write_out, set_stop_time, step
until TstopReached / RootFound, write_out after every step.  With the deterministic pow on both sides every time and every state bit agree."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


@pytest.fixture
def det_pow(O):
    O.set_det_pow(True)
    yield
    O.set_det_pow(False)


def reference_solve(O, model, p, t_final, method=None, **kw):
    """method.rs:881-961 for a batch (p: [nbatch, np]) without a reset operator: (times, states [ncols, nbatch, n], root (t, idx) or None)"""
    if method is not None:
        kw = dict(kw, method=method)
    s = O.OracleSolver(model, p, nbatch=p.shape[0], **kw)
    st = s.state()
    ts, ys = [st["t"]], [st["y"].copy()]
    s.set_stop_time(t_final)
    root = None
    while True:
        r = s.step()
        if r == 1:  # RootFound: state_mut_back(t_root) — the interpolated state at the root — then write_out
            t_root, idx = s.root_info()
            ts.append(t_root); ys.append(s.interpolate(t_root).copy())
            root = (t_root, idx)
            break
        st = s.state()
        ts.append(st["t"]); ys.append(st["y"].copy())
        if r == 2:
            break
    return np.array(ts), np.array(ys), root


def robertson_p(rng, nb):
    return np.stack([0.04 * 2 ** rng.uniform(-1, 1, nb), 1e4 * 2 ** rng.uniform(-1, 1, nb), 3e7 * 2 ** rng.uniform(-1, 1, nb)], axis=1)


def test_every_accepted_step_of_every_member_equals_the_reference_loop_bit_for_bit(H, O, det_pow):
    rng = np.random.default_rng(2026)
    nb = 300
    p = robertson_p(rng, nb)
    kw = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **kw)
    y, t, m, tot = s.solve_adaptive(4e3, max_cols=400, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all() and (m["ncols"] <= 400).all()
    assert (m["ncols"] == m["stats"][0] + 1).all()  # one column per accepted step + the initial state
    for b in list(range(0, nb, 23)) + [nb - 1]:
        ts, ys, root = reference_solve(O, O.MODEL_ROBERTSON_ODE, p[b:b + 1], 4e3, model_size=1, **kw)
        nc = m["ncols"][b]
        assert nc == len(ts) and root is None
        assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0]), f"member {b}"
        assert t[0, b] == 0.0 and t[nc - 1, b] == 4e3  # (columns behind ncols[b] are unspecified)
    # the columns of solve are the columns solve_dense interpolates between: same counters as the save-point run to the same stop time
    _, tot_d = s.solve_dense_adaptive([4e3], group=1)
    assert tot_d["number_of_steps"] == tot["number_of_steps"] and tot_d["number_of_nonlinear_solver_iterations"] == tot["number_of_nonlinear_solver_iterations"]


def test_lock_step_groups_share_their_columns_like_one_batched_solve(H, O, det_pow):
    rng = np.random.default_rng(7)
    nb = 64 + 37  # a full wavefront and a ragged one
    p = robertson_p(rng, nb)
    kw = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **kw)
    y, t, m, tot = s.solve_adaptive(40.0, max_cols=200, group=64)
    assert tot["failed_members"] == 0
    for lo, hi in ((0, 64), (64, nb)):
        ts, ys, _ = reference_solve(O, O.MODEL_ROBERTSON_ODE, p[lo:hi], 40.0, model_size=1, **kw)
        nc = len(ts)
        assert (m["ncols"][lo:hi] == nc).all()
        assert np.array_equal(t[:nc, lo:hi], np.repeat(ts[:, None], hi - lo, axis=1)) and np.array_equal(y[:nc, lo:hi], ys)


def test_too_little_room_is_reported_and_events_end_a_members_columns_at_its_root(H, O, det_pow):
    rng = np.random.default_rng(3)
    nb = 70
    p = robertson_p(rng, nb)
    kw = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **kw)
    y, t, m, _ = s.solve_adaptive(4e3, max_cols=50, group=1)
    assert (m["ncols"] > 50).all()  # counted, not stored
    y2, t2, m2, _ = s.solve_adaptive(4e3, max_cols=int(m["ncols"].max()), group=1)
    assert np.array_equal(m2["ncols"], m["ncols"]) and np.array_equal(y2[:50], y) and np.array_equal(t2[:50], t)
    # a model with a root function: dy/dt = y (1 - y / k), stop at y = 0.5 k
    from diffsol_amd import diffsl as fe
    import diffsl_models as D
    LOGISTIC = "in = [r, k]\nr { 1 }\nk { 1 }\nu_i { y = 0.1 }\nF_i { (r * y) * (1 - (y / k)) }\nstop_i { y - 0.5 * k }\n"
    mdl, mid = fe.DiffslModel(LOGISTIC), D.host_model(O, LOGISTIC)
    pl = np.stack([rng.uniform(0.5, 2.0, nb), rng.uniform(0.8, 1.6, nb)], axis=1)
    tol = dict(rtol=1e-6, atol=[1e-8])
    sl = H.Solver(mdl, pl, nbatch=nb, **tol)
    y, t, m, tot = sl.solve_adaptive(50.0, max_cols=300, group=1)
    assert tot["failed_members"] == 0 and (m["root_idx"] == 0).all()
    for b in range(0, nb, 9):
        ts, ys, root = reference_solve(O, mid, pl[b:b + 1], 50.0, **tol)
        nc = m["ncols"][b]
        assert root is not None and nc == len(ts) and m["t_root"][b] == root[0] == t[nc - 1, b]
        assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0])
        assert abs(y[nc - 1, b, 0] - 0.5 * pl[b, 1]) < 1e-6


@pytest.mark.parametrize("model,size,t_final,group", [("heat1d", 20, 0.05, 1), ("spm", 20, 1200.0, 1), ("heat1d", 64, 0.02, 64)])
def test_banded_lane_per_member_form_returns_every_accepted_step(H, O, det_pow, model, size, t_final, group):
    """VERDICT r4 missing 3: OdeSolverMethod::solve outside the register-resident forms.  k_bdf_lane_banded (state in per-lane memory, banded LU; BASELINE config 4's
    kernel) writes the state after every accepted step: times and states equal the reference's loop over the oracle's stepping solver, bit for bit — per member, and per
    64-member lock-step group (the single-particle model's voltage cut-offs end a member's columns at its root)."""
    from helpers import ORACLE_MODEL
    rng = np.random.default_rng(size + group)
    nb = 70 if group == 1 else 64 + 9
    p = rng.uniform(0.6, 1.4, (nb, 1))
    tol = dict(rtol=1e-6, atol=[1e-6])
    s = H.Solver(model, p, nbatch=nb, model_size=size, **tol)
    y, t, m, tot = s.solve_adaptive(t_final, max_cols=400, group=group)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all() and (m["ncols"] <= 400).all()
    if group == 1:
        for b in list(range(0, nb, 13)) + [nb - 1]:
            ts, ys, root = reference_solve(O, ORACLE_MODEL[model], p[b:b + 1], t_final, model_size=size, **tol)
            nc = m["ncols"][b]
            assert nc == len(ts), (b, nc, len(ts))
            assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0]), f"member {b}"
            assert (root is None) == (m["root_idx"][b] < 0)
            if root is not None:
                assert m["t_root"][b] == root[0] == t[nc - 1, b]
    else:
        for lo, hi in ((0, 64), (64, nb)):
            ts, ys, _ = reference_solve(O, ORACLE_MODEL[model], p[lo:hi], t_final, model_size=size, **tol)
            nc = len(ts)
            assert (m["ncols"][lo:hi] == nc).all()
            assert np.array_equal(t[:nc, lo:hi], np.repeat(ts[:, None], hi - lo, axis=1)) and np.array_equal(y[:nc, lo:hi], ys)
    # the columns of solve are the steps solve_dense interpolates between
    _, tot_d = s.solve_dense_adaptive([t_final], group=group)
    assert tot_d["number_of_steps"] == tot["number_of_steps"]


@pytest.mark.parametrize("n", [30, 100, 200])
def test_wavefront_and_workgroup_per_member_forms_return_every_accepted_step(H, O, det_pow, n):
    """k_bdf_wave_member (n <= 64: a wavefront per member), k_bdf_team_member (a workgroup per member; n = 200 with the factors in global scratch): dense models,
    per-member control, every accepted step out — times and states equal the reference's loop over the oracle's stepping solver."""
    from helpers import ORACLE_MODEL
    rng = np.random.default_rng(n)
    nb = 11
    p = rng.uniform(0.5, 2.0, (nb, n))
    tol = dict(rtol=1e-6, atol=[1e-6])
    s = H.Solver("gaussian_decay", p, nbatch=nb, model_size=n, **tol)
    y, t, m, tot = s.solve_adaptive(1.5, max_cols=300, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all() and (m["ncols"] <= 300).all()
    for b in (0, nb // 2, nb - 1):
        ts, ys, root = reference_solve(O, ORACLE_MODEL["gaussian_decay"], p[b:b + 1], 1.5, model_size=n, **tol)
        nc = m["ncols"][b]
        assert nc == len(ts) and root is None
        assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0]), f"member {b}"


@pytest.mark.parametrize("method", ["tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("n", [30, 100, 200])
def test_wavefront_and_workgroup_per_member_sdirk_return_every_accepted_step(H, O, det_pow, method, n):
    """OdeSolverMethod::solve inside k_sdirk_wave_member (a wavefront per member for n <= 64, a workgroup per member beyond; n = 200 with the factors in global scratch)
    for TR-BDF2 / ESDIRK34 (dsh_sdirk_solve_wave_member_steps): times and states of every accepted step equal the reference's loop over the oracle's stepping solver."""
    from helpers import ORACLE_MODEL
    hm = {"tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
    om = {"tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
    rng = np.random.default_rng(n + len(method))
    nb = 11
    p = rng.uniform(0.5, 2.0, (nb, n))
    tol = dict(rtol=1e-6, atol=[1e-6])
    s = H.Solver("gaussian_decay", p, nbatch=nb, model_size=n, method=hm, **tol)
    y, t, m, tot = s.solve_adaptive(1.5, max_cols=400, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all() and (m["ncols"] <= 400).all()
    for b in (0, nb // 2, nb - 1):
        ts, ys, root = reference_solve(O, ORACLE_MODEL["gaussian_decay"], p[b:b + 1], 1.5, model_size=n, method=om, **tol)
        nc = m["ncols"][b]
        assert nc == len(ts) and root is None
        assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0]), f"member {b}"
    _, tot_d = s.solve_dense_adaptive([1.5], group=1)
    assert tot_d["number_of_steps"] == tot["number_of_steps"]


@pytest.mark.parametrize("method", ["tr_bdf2", "esdirk34"])
@pytest.mark.parametrize("case", ["robertson", "logistic_root", "heat20"])
def test_resident_sdirk_kernels_return_every_accepted_step(H, O, det_pow, method, case):
    """OdeSolverMethod::solve for TR-BDF2 / ESDIRK34 inside the launch of k_sdirk_resident (register-resident models, and the banded lane-per-member form of heat1d):
    times and states of every accepted step equal the reference's loop over the oracle's stepping solver; an event ends a member's columns at its root."""
    from helpers import ORACLE_MODEL
    hm = {"tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
    om = {"tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
    rng = np.random.default_rng(len(case) + len(method))
    nb = 70
    if case == "robertson":
        p, t_final, kw, hmodel, omodel, size = robertson_p(rng, nb), 40.0, dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6]), "robertson_ode", O.MODEL_ROBERTSON_ODE, 1
    elif case == "heat20":
        p, t_final, kw, hmodel, omodel, size = rng.uniform(0.6, 1.4, (nb, 1)), 0.05, dict(rtol=1e-6, atol=[1e-6]), "heat1d", ORACLE_MODEL["heat1d"], 20
    else:
        from diffsol_amd import diffsl as fe
        import diffsl_models as D
        LOGISTIC = "in = [r, k]\nr { 1 }\nk { 1 }\nu_i { y = 0.1 }\nF_i { (r * y) * (1 - (y / k)) }\nstop_i { y - 0.5 * k }\n"
        p, t_final, kw, size = np.stack([rng.uniform(0.5, 2.0, nb), rng.uniform(0.8, 1.2, nb)], axis=1), 50.0, dict(rtol=1e-6, atol=[1e-8]), 0
        hmodel, omodel = fe.DiffslModel(LOGISTIC), D.host_model(O, LOGISTIC)
    skw = dict(kw) if size == 0 else dict(kw, model_size=size)
    s = H.Solver(hmodel, p, nbatch=nb, method=hm, **skw)
    y, t, m, tot = s.solve_adaptive(t_final, max_cols=500, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all() and (m["ncols"] <= 500).all()
    for b in list(range(0, nb, 17)) + [nb - 1]:
        ts, ys, root = reference_solve(O, omodel, p[b:b + 1], t_final, method=om, **skw)
        nc = m["ncols"][b]
        assert nc == len(ts), (b, nc, len(ts))
        assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0]), f"member {b}"
        assert (root is None) == (m["root_idx"][b] < 0)
        if root is not None:
            assert m["t_root"][b] == root[0] == t[nc - 1, b]
    if case == "logistic_root":
        assert (m["root_idx"] == 0).all()


DAE6 = ("in = [k]\nk { 1 }\nu_i { a = 1, b = 1, c = 1, d = 1, e = 1, z = 5 }\ndudt_i { da = 0, db = 0, dc = 0, dd = 0, de = 0, dz = 0 }\n"
        "M_i { da, db, dc, dd, de, 0 }\nF_i { -k * a, -2 * k * b, -3 * k * c, -4 * k * d, -5 * k * e, z - (a + b + c + d + e) }\n")


@pytest.mark.parametrize("method", ["bdf", "tr_bdf2", "esdirk34"])
def test_wavefront_per_member_dae_returns_every_accepted_step(H, O, det_pow, method):
    """Five decays and one algebraic sum (n = 6, a mass matrix, dense: served by the wavefront-per-member kernels through the run-time-sized twin): every accepted step
    equals the reference's loop over the oracle's stepping solver."""
    from diffsol_amd import diffsl as fe
    import diffsl_models as D
    hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
    om = {"bdf": None, "tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
    rng = np.random.default_rng(3)
    nb = 9
    p = rng.uniform(0.5, 2.0, (nb, 1))
    tol = dict(rtol=1e-6, atol=[1e-8])
    s = H.Solver(fe.DiffslModel(DAE6), p, nbatch=nb, method=hm, **tol)
    y, t, m, tot = s.solve_adaptive(1.0, max_cols=400, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all()
    mid = D.host_model(O, DAE6)
    for b in (0, 4, nb - 1):
        ts, ys, root = reference_solve(O, mid, p[b:b + 1], 1.0, method=om, **tol)
        nc = m["ncols"][b]
        assert nc == len(ts) and root is None
        assert np.array_equal(t[:nc, b], ts) and np.array_equal(y[:nc, b], ys[:, 0]), f"member {b}"
        assert np.max(np.abs(y[:nc, b, 5] - y[:nc, b, :5].sum(axis=1))) < 1e-4  # the algebraic row, to the tolerances
