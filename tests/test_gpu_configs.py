"""BASELINE.json configs[2..4] at the shapes BASELINE states ("config 3 is not exercised at its shape in -m gpu"; configs 4 and 5 only at
9-70 members).  Each config: (i) a subset of the full-size ensemble bit for bit against the CPU oracle through the same integrator at the same n,
(ii) the full-size run with size-independent properties as assertions (closed forms, invariants, monotonicity, event bookkeeping), and sampled
members against their own oracle solves.

  C3  heat1d (examples/pde-heat) n = 512 x 4096, TR-BDF2, both LU structures (banded kernels for the tridiagonal operand / dense blocked LU)
  C4  single-particle battery model n = 42, BDF, per-member voltage cut-offs, one GPU's shard of the 8-GPU ensemble (32 768 members)
  C5  series RLC DAE n = 4 x 65 536, ESDIRK34, per-member events
"""
import numpy as np
import pytest

from helpers import ORACLE_MODEL

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


def heat_fourier(n, D, t, terms=200):
    """Fourier series of the triangle initial condition (test_models/heat1d.rs:62-96): u(x,t) = 8/pi^2 sum sin((2m-1) pi x) exp(-(2m-1)^2 pi^2 D t)/(2m-1)^2 (alternating sign)."""
    x = (np.arange(n) + 1) / (n + 1)
    m = np.arange(1, terms)[:, None, None]
    k = 2 * m - 1
    sign = np.where(m % 2 == 1, 1.0, -1.0)
    return (sign * np.sin(k * np.pi * x[None, None, :]) * np.exp(-k ** 2 * np.pi ** 2 * np.asarray(D)[None, :, None] * t) / k ** 2).sum(0) * 8 / np.pi ** 2


def heat_params(nb):
    return np.random.default_rng(12345).uniform(0.5, 2.0, nb)  # D_b ~ U[0.5, 2]


@pytest.mark.parametrize("structure", ["auto", "dense"])
def test_config3_heat1d_n512_tr_bdf2_members_equal_the_oracle_bitwise(H, O, monkeypatch, structure):
    """n = 512 through the INTEGRATOR (not only the LU): 8 members of the C3 sweep, TR-BDF2, rtol = atol = 1e-6, to t = 0.5 — states at two times and all
    counters bit for bit against the oracle's lock-step run; with the banded LU kernels (tridiagonal operand found / declared) and with the dense blocked LU."""
    monkeypatch.setenv("DSH_LU_STRUCTURE", structure)
    D = heat_params(4096)[:8]
    kw = dict(nbatch=8, model_size=512, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    s = H.Solver("heat1d", D[:, None], **kw)
    assert s.n == 512 and not s.fused
    times = [0.01, 0.5]
    y, _ = s.solve_to_points(times)
    o = O.OracleSolver(ORACLE_MODEL["heat1d"], D[:, None], **kw)
    yo, _ = o.solve_to_points(times)
    assert np.array_equal(y, yo) and s.stats() == o.stats()
    assert np.abs(y[1] - heat_fourier(512, D, 0.5)).max() < 2e-5


def test_config3_dense_mode_on_the_default_matrix_core_lu_stays_within_rounding_of_the_oracle(H, O, monkeypatch):
    """BASELINE configs[2] says "banded-as-dense LU": with DSH_LU_STRUCTURE=dense and the library's DEFAULT dense LU for n = 512 (the matrix-core kernel of
    dsh_lu_tiled.hpp; every other test of this file pins the exact mode) the integrator must take the oracle's step sequence — all counters equal — and
    return its states to 1e-9 relative (north_star: 1e-6): the factors differ from the exact ones in the last bits only."""
    monkeypatch.delenv("DSH_LU_EXACT", raising=False)
    monkeypatch.setenv("DSH_LU_STRUCTURE", "dense")
    D = heat_params(4096)[:8]
    kw = dict(nbatch=8, model_size=512, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    s = H.Solver("heat1d", D[:, None], **kw)
    times = [0.01, 0.5]
    y, _ = s.solve_to_points(times)
    o = O.OracleSolver(ORACLE_MODEL["heat1d"], D[:, None], **kw)
    yo, _ = o.solve_to_points(times)
    assert s.stats() == o.stats()
    assert np.max(np.abs(y - yo)) <= 1e-9 * np.max(np.abs(yo))
    assert np.abs(y[1] - heat_fourier(512, D, 0.5)).max() < 2e-5


def test_config3_dense_mode_64_members_default_lu_against_the_exact_lu(H, monkeypatch):
    """the DEFAULT dense LU is the kernel bench.py's `c3_dense` row times, and 8 members were thin evidence for it.  64 members of the C3 sweep,
    n = 512, TR-BDF2 to t = 0.5 with the default matrix-core LU against the same run with the exact blocked LU (DSH_LU_EXACT=1: bit-identical to the oracle — the test
    above for 8 members, the banded == dense comparison below at full size): every counter equal (the same step sequence), states within 1e-9 relative."""
    monkeypatch.setenv("DSH_LU_STRUCTURE", "dense")
    D = heat_params(4096)[:64]
    kw = dict(nbatch=64, model_size=512, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    times = [0.01, 0.1, 0.5]
    monkeypatch.setenv("DSH_LU_EXACT", "1")
    se = H.Solver("heat1d", D[:, None], **kw)
    ye, _ = se.solve_to_points(times)
    monkeypatch.delenv("DSH_LU_EXACT", raising=False)
    sd = H.Solver("heat1d", D[:, None], **kw)
    yd, _ = sd.solve_to_points(times)
    assert sd.stats() == se.stats()
    assert np.max(np.abs(yd - ye)) <= 1e-9 * np.max(np.abs(ye)) and not np.array_equal(yd, ye)
    assert np.abs(yd[2] - heat_fourier(512, D, 0.5)).max() < 2e-5


@pytest.fixture(scope="module")
def config3_full(H):
    out = {}
    D = heat_params(4096)
    import os
    for structure in ("auto", "dense"):
        os.environ["DSH_LU_STRUCTURE"] = structure
        try:
            s = H.Solver("heat1d", D[:, None], nbatch=4096, model_size=512, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
            y, _ = s.solve_to_points([0.5])
            out[structure] = (y[0], s.stats())
        finally:
            os.environ.pop("DSH_LU_STRUCTURE", None)
    return D, out


def test_config3_full_size_heat1d_512x4096_against_the_fourier_series(config3_full):
    """The whole of BASELINE configs[2]: 4096 members x 512 states, TR-BDF2 to t = 0.5.  EVERY member against the Fourier series (the reference's own
    known-answer check, tolerance scaled to rtol = atol = 1e-6), positivity, symmetry of the profile, decay ordered by diffusivity; banded and dense
    LU paths give the same bits and the same counters."""
    D, out = config3_full
    y, st = out["auto"]
    assert y.shape == (4096, 512) and np.isfinite(y).all()
    ref = heat_fourier(512, D, 0.5)
    assert np.abs(y - ref).max() < 2e-5
    assert y.min() > -1e-6 and np.abs(y - y[:, ::-1]).max() < 1e-9          # u >= 0, symmetric about x = 1/2
    order = np.argsort(D)
    assert np.all(np.diff(y[order, 256]) < 1e-9)                               # larger diffusivity -> smaller mid-point value
    assert st["number_of_steps"] > 100 and st["number_of_nonlinear_solver_fails"] == 0
    yd, std = out["dense"]
    assert np.array_equal(y, yd) and st == std


def test_config3_model_at_65536_members_runs_host_driven_because_its_matrices_are_band_containers(H):
    """heat1d n = 512 x 65 536 members, host-driven TR-BDF2.  Jacobian, mass, M - cJ and the LU factors of this declared-band model live in
    band containers (3 n, resp. 4 n doubles per member): 0.8 - 1.1 GB each at this size, where the dense containers would take 137 GB each — the run would not
    fit the device.  Checks: it runs, the device memory in use stays below 24 GB, the solution is the Fourier series' to the integration tolerance."""
    import torch
    nb, n = 65536, 512
    D = np.random.default_rng(12345).uniform(0.5, 2.0, nb)
    free0, _ = torch.cuda.mem_get_info()
    s = H.Solver("heat1d", D[:, None], nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    y, _ = s.solve_to_points([0.05])
    free1, _ = torch.cuda.mem_get_info()
    assert (free0 - free1) < 24 * 2**30, (free0 - free1) / 2**30
    assert np.isfinite(y).all() and s.stats()["number_of_steps"] > 20
    pick = np.arange(0, nb, 1024)
    assert np.abs(y[0, pick] - heat_fourier(n, D[pick], 0.05)).max() < 2e-4  # the semi-discretisation error at this early time dominates


def spm_currents(nb):
    return np.random.default_rng(12345).uniform(0.6, 1.4, nb)  # I_b ~ U[0.6, 1.4] A


def test_config4_spm_32768_members_with_voltage_cutoffs(H, O, det_pow):
    """One GPU's shard of BASELINE configs[3] (262 144 = 8 x 32 768): single-particle model n = 42, BDF, one-hour discharge with the stop conditions armed,
    every member its own step sizes and its own cut-off time (device-resident, one lane per member, banded LU).  Properties over all 32 768 members:
    discharge capacity = I t at every save point before the member's stop, cut-off time decreasing in the current, NaN after the stop, status 0;
    a random sample of members bit for bit (states, counters, event times) against their own oracle solves."""
    nb = 32768
    cur = spm_currents(nb)
    s = H.Solver("spm", cur[:, None], nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    assert s.n == 42
    t_eval = np.linspace(360.0, 3600.0, 10)
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all()
    hit = m["root_idx"] >= 0
    assert 0.8 * nb < hit.sum() < nb                                           # most members reach 3.105 V within the hour (I > ~0.66 A)
    assert np.all(m["t_root"][hit] > 1500.0) and np.all(m["t_root"][hit] <= 3600.0)
    o = np.argsort(cur[hit])
    assert np.all(np.diff(m["t_root"][hit][o]) <= 0.0)                          # higher current -> earlier cut-off
    for k, t in enumerate(t_eval):
        live = (~hit) | (m["t_root"] >= t)                                     # column k is a regular save point of the member (not past its stop)
        assert np.allclose(y[k, live, 0], cur[live] * t / 3600.0, rtol=1e-5)   # first state = discharge capacity [Ah]
        dead = hit & (m["ncols"] <= k)
        assert np.isnan(y[k, dead]).all()
    rng = np.random.default_rng(3)
    pick = np.sort(rng.choice(nb, 24, replace=False))
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL["spm"], cur[pick, None], t_eval, model_size=20, nthreads=8, rtol=1e-6, atol=[1e-6])
    ref = O.solve_dense_independent.last_roots
    assert failed == 0 and np.array_equal(y[:, pick], np.transpose(yo, (1, 0, 2)), equal_nan=True)
    assert np.array_equal(m["stats"].T[pick], so) and np.array_equal(m["t_root"][pick], ref["t_root"], equal_nan=True)
    assert np.array_equal(m["root_idx"][pick], ref["root_idx"]) and np.array_equal(m["ncols"][pick], ref["ncols"])


@pytest.mark.parametrize("dae", [False, True])
def test_config4_whole_262144_member_ensemble_on_one_gpu_equals_its_shards_and_the_oracle(H, O, det_pow, dae):
    """bench.py's `c4_*_262144` rows (the whole 8-GPU ensemble of BASELINE configs[3] on one GPU) were guarded by `finite` / `failed_members` only.
    The 262 144-member launch — the DEFAULT code object of the banded lane BDF; every smaller ensemble of this tier runs the small-ensemble one — must give, member for
    member, the bits of the 32 768-member shard launch (members are independent problems under per-member control: rank 0's shard of the 8-GPU job) and of the oracle
    on a random sample drawn from the whole ensemble (states, counters, stop times); capacity = I t at every live save point."""
    import diffsl_models as DM
    from diffsol_amd import diffsl as fe
    nb, shard = 262144, 32768
    cur = spm_currents(nb)
    t_eval = np.linspace(360.0, 3600.0, 10)
    code = DM.spm_dae(20) if dae else None
    mk = (lambda c: H.Solver(fe.DiffslModel(code), c[:, None], nbatch=len(c), rtol=1e-6, atol=[1e-6])) if dae else \
         (lambda c: H.Solver("spm", c[:, None], nbatch=len(c), model_size=20, rtol=1e-6, atol=[1e-6]))
    y, tot, m = mk(cur).solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all()
    hit = m["root_idx"] >= 0
    for k, t in enumerate(t_eval):
        live = (~hit) | (m["t_root"] >= t)
        assert np.allclose(y[k, live, 0], cur[live] * t / 3600.0, rtol=1e-5)
    ys, _, ms = mk(cur[:shard]).solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert np.array_equal(y[:, :shard], ys, equal_nan=True) and np.array_equal(m["stats"][:, :shard], ms["stats"])
    assert np.array_equal(m["t_root"][:shard], ms["t_root"], equal_nan=True) and np.array_equal(m["ncols"][:shard], ms["ncols"])
    pick = np.sort(np.random.default_rng(8).choice(nb, 16, replace=False))
    oid = DM.host_model(O, code) if dae else ORACLE_MODEL["spm"]
    kw = {} if dae else {"model_size": 20}
    yo, so, failed = O.solve_dense_independent(oid, cur[pick, None], t_eval, nthreads=8, rtol=1e-6, atol=[1e-6], **kw)
    ref = O.solve_dense_independent.last_roots
    assert failed == 0 and np.array_equal(y[:, pick], np.transpose(yo, (1, 0, 2)), equal_nan=True)
    assert np.array_equal(m["stats"].T[pick], so) and np.array_equal(m["t_root"][pick], ref["t_root"], equal_nan=True) and np.array_equal(m["ncols"][pick], ref["ncols"])


def test_config4_as_worded_singular_mass_spm_dae_32768_members(H, O, det_pow):
    """BASELINE configs[3] in its own words — "SPM DAE (singular mass matrix)": the battery model with the terminal voltage as an ALGEBRAIC state
    (tests/diffsl_models.py spm_dae(20): n = 43, M = diag(1.., 0, ..1), bandwidth 2), one GPU's shard of 32 768 members on the lane-per-member banded BDF
    (k_bdf_lane_banded with the diagonal mass matrix and the consistent initialisation per lane).  Same properties as the identity-mass run
    above — capacity = I t, cut-off time decreasing in the current, NaN after the stop — plus: V at the first save point sits on its constraint, and the
    cut-off times agree with the identity-mass formulation's (same physics, the voltage an expression there) to the integration tolerance.  24 sampled
    members bit for bit (states, counters, stop times) against their own oracle solves of the DiffSL host twin."""
    import diffsl_models as DM
    from diffsol_amd import diffsl as fe
    nb = 32768
    cur = spm_currents(nb)
    code = DM.spm_dae(20)
    model = fe.DiffslModel(code)
    assert model.n == 43 and model.has_mass and model.lane_model_id is not None
    s = H.Solver(model, cur[:, None], nbatch=nb, rtol=1e-6, atol=[1e-6])
    t_eval = np.linspace(360.0, 3600.0, 10)
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all()
    hit = m["root_idx"] >= 0
    assert 0.8 * nb < hit.sum() < nb and np.all(m["root_idx"][hit] == 0)        # stop 0: V - 3.105
    assert np.all(m["t_root"][hit] > 1500.0) and np.all(m["t_root"][hit] <= 3600.0)
    o = np.argsort(cur[hit])
    assert np.all(np.diff(m["t_root"][hit][o]) <= 0.0)
    for k, t in enumerate(t_eval):
        live = (~hit) | (m["t_root"] >= t)
        assert np.allclose(y[k, live, 0], cur[live] * t / 3600.0, rtol=1e-5)
        assert np.all(y[k, live, 22] > 3.105 - 1e-6) and np.all(y[k, live, 22] < 4.1)  # the algebraic state between its two stops
        assert np.isnan(y[k, hit & (m["ncols"] <= k)]).all()
    ident = H.Solver("spm", cur[:, None], nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    _, _, mi = ident.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    both = hit & (mi["root_idx"] >= 0)
    assert both.sum() > 0.8 * nb and np.abs(m["t_root"][both] / mi["t_root"][both] - 1.0).max() < 2e-4
    pick = np.sort(np.random.default_rng(3).choice(nb, 24, replace=False))
    yo, so, failed = O.solve_dense_independent(DM.host_model(O, code), cur[pick, None], t_eval, nthreads=8, rtol=1e-6, atol=[1e-6])
    ref = O.solve_dense_independent.last_roots
    assert failed == 0 and np.array_equal(y[:, pick], np.transpose(yo, (1, 0, 2)), equal_nan=True)
    assert np.array_equal(m["stats"].T[pick], so) and np.array_equal(m["t_root"][pick], ref["t_root"], equal_nan=True)
    assert np.array_equal(m["root_idx"][pick], ref["root_idx"]) and np.array_equal(m["ncols"][pick], ref["ncols"])


def rlc_params(nb, thresh):
    rng = np.random.default_rng(12345)  # R ~ U[50, 200], L = 1, C ~ logU[5e-4, 2e-3], V0 = 10, omega = 100
    R = rng.uniform(50.0, 200.0, nb)
    Cc = np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    return np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, thresh)], axis=1)


@pytest.mark.parametrize("group", [1, 64])
def test_config5_fast_arithmetic_build_makes_the_decisions_of_the_exact_kernel_at_full_size(H, group):
    """Round 6: the fast-arithmetic build of the device-resident ESDIRK34 (dsh_sdirk_fast.hip; `deterministic_pow = 2`, the library default of solve_dense since this
    round) on BASELINE configs[4] at its full 65 536 members against the exact kernel (bit-identical to the oracle: the test below): every member's five counters,
    its number of output columns and the index of its event are equal — the same step-size, refactorisation and event decisions in every wavefront —, states within 1e-9 relative + 1e-11 and event
    times within 1e-9 relative (north_star: 1e-6)."""
    nb = 65536
    p = rlc_params(nb, 0.03 if group == 1 else 1e3)
    s = H.Solver("rlc", p, nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
    t_eval = np.linspace(0.005, 1.0, 12)
    ye, tote, me = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group, deterministic_pow=1)
    yf, totf, mf = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group, deterministic_pow=2)
    assert tote["failed_members"] == 0 and totf["failed_members"] == 0
    assert np.array_equal(me["stats"], mf["stats"]) and np.array_equal(me["ncols"], mf["ncols"]) and np.array_equal(me["root_idx"], mf["root_idx"])
    hit = me["root_idx"] >= 0
    if group == 1:
        assert hit.sum() > 0.3 * nb
        assert np.max(np.abs(mf["t_root"][hit] - me["t_root"][hit]) / me["t_root"][hit]) < 1e-9
    live = np.isfinite(ye)
    assert np.array_equal(live, np.isfinite(yf))
    assert np.all(np.abs(yf - ye)[live] <= 1e-9 * np.abs(ye)[live] + 1e-11)  # 1e-9 relative, with a floor five decades below atol for the oscillating currents' zero crossings (measured: 2.3e-12)
    assert not np.array_equal(yf[live], ye[live])  # it IS another arithmetic


def test_config5_rlc_65536_members_esdirk34_with_per_member_events(H, O, det_pow):
    """BASELINE configs[4] at full size: 65 536 series-RLC DAEs (singular mass), ESDIRK34, t in [0, 1], root iR - 0.03 A armed: every member stops at
    its own crossing (what the reference's batched root finding cannot do).  Properties over all members: the algebraic equations of the DAE hold at
    every save point, the state at the root sits on the event surface, event bookkeeping is consistent; a sample of members bit for bit against their
    own oracle solves.  The event-free run (threshold out of reach) in wavefront lock-step groups reaches t = 1 for every member."""
    nb = 65536
    p = rlc_params(nb, 0.03)
    s = H.Solver("rlc", p, nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
    t_eval = np.linspace(0.005, 1.0, 12)
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    assert tot["failed_members"] == 0 and (m["status"] == 0).all()
    hit = m["root_idx"] >= 0
    assert 0.3 * nb < hit.sum() < nb and np.all(m["root_idx"][hit] == 0)
    assert np.all(m["t_root"][hit] > 0.0) and np.all(m["t_root"][hit] < 1.0) and np.isnan(m["t_root"][~hit]).all()
    # states: (iR, iL, iC, V) of examples/electrical-circuits: 0 = V - R iR (row 0) and 0 = iL - iR - iC (row 2) are the algebraic equations (M = diag(0,1,0,1))
    R = p[:, 0]
    for k in range(len(t_eval)):
        live = m["ncols"] > k
        yk = y[k, live]
        assert np.isfinite(yk).all()
        assert np.abs(yk[:, 3] / R[live] - yk[:, 0]).max() < 2e-5 and np.abs(yk[:, 1] - yk[:, 0] - yk[:, 2]).max() < 2e-5
        assert np.isnan(y[k, ~live]).all()
    # the column after a member's last regular save point holds the state AT the root: iR = 0.03 there
    rows = np.nonzero(hit)[0]
    at_root = y[m["ncols"][rows] - 1, rows]
    assert np.abs(at_root[:, 0] - 0.03).max() < 1e-6
    pick = np.sort(np.random.default_rng(4).choice(nb, 48, replace=False))
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL["rlc"], p[pick], t_eval, model_size=1, nthreads=8, method=2, rtol=1e-6, atol=[1e-6])
    ref = O.solve_dense_independent.last_roots
    assert failed == 0 and np.array_equal(y[:, pick], np.transpose(yo, (1, 0, 2)), equal_nan=True) and np.array_equal(m["stats"].T[pick], so)
    assert np.array_equal(m["t_root"][pick], ref["t_root"], equal_nan=True) and np.array_equal(m["ncols"][pick], ref["ncols"])
    # event-free, wavefront lock-step groups (the default route of solve_dense for a model whose roots cannot fire is per member; force groups here)
    p2 = rlc_params(nb, 1e3)
    s2 = H.Solver("rlc", p2, nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
    y2, tot2 = s2.solve_dense_adaptive([0.5, 1.0], group=64)
    assert tot2["failed_members"] == 0 and np.isfinite(y2).all()
    assert np.abs(y2[1][:, 3] / R - y2[1][:, 0]).max() < 2e-5
    g0 = slice(0, 64)
    yo2, _, failed2 = O.solve_dense_independent(ORACLE_MODEL["rlc"], p2[g0], [0.5, 1.0], model_size=1, nthreads=1, group=64, method=2, rtol=1e-6, atol=[1e-6])
    assert failed2 == 0 and np.array_equal(y2[:, g0], np.transpose(yo2, (1, 0, 2)))
