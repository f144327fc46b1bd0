"""Banded (non-diagonal) mass matrices in the lane-per-member BDF (VERDICT r3 missing 3; op/bdf.rs:240-256 / :273-300, op/init.rs:31-64): run-time-sized DiffSL models
whose mass matrix is tridiagonal — a finite-element heat equation (consistent mass matrix tridiag(1/6, 4/6, 1/6)) and the same with algebraic boundary rows (a
singular banded mass matrix: consistent initialisation on -M_u restricted to the differential rows and columns).  k_bdf_lane_banded keeps M's band next to the
Jacobian's; counters and every output bit equal the oracle's per-member solve_dense on the generated host twin."""
import numpy as np
import pytest

import diffsl_models as D

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


def tri(a, b, c, n, name, rows=None):
    rows = range(n) if rows is None else rows
    ent = []
    for i in rows:
        if i > 0:
            ent.append(f"  ({i},{i - 1}): {a!r}")
        ent.append(f"  ({i},{i}): {b!r}")
        if i + 1 < n:
            ent.append(f"  ({i},{i + 1}): {c!r}")
    if n - 1 not in rows:
        ent.append(f"  ({n - 1},{n - 1}): 0.0")  # the tensor's extent
    return f"{name}_ij {{\n" + ",\n".join(ent) + "\n}\n"


def fem_heat(n):
    """linear finite elements on n interior nodes: M du/dt = d K u, M = h tridiag(1/6, 4/6, 1/6), K = tridiag(1, -2, 1) / h, hat initial condition"""
    h = 1.0 / (n + 1)
    ic = ",\n".join(f"  ({i}): {2.0 * (i + 1) * h if (i + 1) * h < 0.5 else 2.0 * (1.0 - (i + 1) * h)!r}" for i in range(n))
    return (f"in = [d]\nd {{ 1.0 }}\n" + tri(1.0 / h, -2.0 / h, 1.0 / h, n, "A") + tri(h / 6, 4 * h / 6, h / 6, n, "B") +
            f"u_i {{\n{ic}\n}}\ndudt_i {{ (0:{n}): 0.0 }}\nlap_i {{ A_ij * u_j }}\nM_i {{ B_ij * dudt_j }}\nF_i {{ d * lap_i }}\n")


def fem_heat_dae(n):
    """the same rod with its two end values kept as ALGEBRAIC states (rows 0 and n - 1 of M are zero, their equations pin the ends to a time-dependent value and to
    zero): the interior rows of the consistent mass matrix still couple to the end columns, so M is a singular banded matrix"""
    h = 1.0 / (n - 1)
    inner = range(1, n - 1)
    ic = ",\n".join(f"  ({i}): {float(np.sin(np.pi * i * h)) + 0.3 * (1 - i * h)!r}" for i in range(n))
    sel0 = "e0_i { (0): 1.0, (1:%d): 0.0 }" % n
    seln = "en_i { (0:%d): 0.0, (%d): 1.0 }" % (n - 1, n - 1)
    return (f"in = [d, g]\nd {{ 1.0 }}\ng {{ 0.3 }}\n" + tri(1.0 / h, -2.0 / h, 1.0 / h, n, "A", inner) + tri(h / 6, 4 * h / 6, h / 6, n, "B", inner) + sel0 + "\n" + seln + "\n" +
            f"u_i {{\n{ic}\n}}\ndudt_i {{ (0:{n}): 0.0 }}\nlap_i {{ A_ij * u_j }}\nM_i {{ B_ij * dudt_j }}\n"
            f"F_i {{ d * lap_i + e0_i * (u_i - g * (1 + 0.5 * sin(3 * t))) + en_i * u_i }}\n")


@pytest.mark.parametrize("which,n,group", [("ode", 12, 1), ("ode", 20, 64), ("dae", 14, 1), ("dae", 14, 64)])
def test_banded_mass_matrices_in_the_lane_per_member_bdf_are_bit_identical_to_the_oracle(H, O, det_pow, which, n, group):
    from diffsol_amd import diffsl as fe
    import diffsol_amd
    code = fem_heat(n) if which == "ode" else fem_heat_dae(n)
    m, mid = fe.DiffslModel(code), D.host_model(O, code)
    assert m.form == fe.FORM_DYNAMIC and m.has_mass and max(m.band[2], m.band[3]) == 1
    dev = diffsol_amd._ffi.load_device_lib()
    assert m.lane_model_id is not None and dev.dsh_model_lane_twin(m.model_id, 0) == m.lane_model_id
    nb = 100
    rng = np.random.default_rng(n + group)
    p = rng.uniform(0.5, 2.0, (nb, 1)) if which == "ode" else np.stack([rng.uniform(0.5, 2.0, nb), rng.uniform(0.1, 0.5, nb)], axis=1)
    t_eval = [0.0, 0.01, 0.05, 0.2, 0.5]
    tol = dict(rtol=1e-6, atol=[1e-8])
    s = H.Solver(m, p, nbatch=nb, **tol)
    y, tot, mm = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
    yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, group=group, **tol)
    assert failed == 0 and tot["failed_members"] == 0 and (mm["status"] == 0).all()
    assert np.array_equal(mm["stats"].T, so), "counters differ"
    assert np.array_equal(y, np.transpose(yo, (1, 0, 2))), "states differ"
    if which == "ode":  # the lowest mode of the semi-discrete problem decays at d * lambda_1, lambda_1 = (6 / h^2) (1 - c) / (2 + c), c = cos(pi h)
        h = 1.0 / (n + 1)
        c = np.cos(np.pi * h)
        lam = 6.0 / h ** 2 * (1 - c) / (2 + c)
        mid_node = n // 2
        ratio = y[4, :, mid_node] / y[3, :, mid_node]
        assert np.allclose(ratio, np.exp(-p[:, 0] * lam * 0.3), rtol=2e-3)
    else:
        assert np.allclose(y[1:, :, 0], p[None, :, 1] * (1 + 0.5 * np.sin(3 * np.array(t_eval[1:])))[:, None], atol=1e-6) and np.abs(y[1:, :, n - 1]).max() < 1e-6
