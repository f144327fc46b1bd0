import numpy as np

ORACLE_MODEL = {"exponential_decay": 0, "exponential_decay_with_algebraic": 1, "exponential_decay_with_algebraic_batched": 2, "robertson_ode": 3,
                "robertson": 4, "dydt_y2": 5, "gaussian_decay": 6, "heat1d": 7, "rlc": 8, "exponential_decay_with_root": 9, "spm": 10,
                "heat2d": 11, "foodweb": 12}
METHOD = {"bdf": 0, "tr_bdf2": 1, "esdirk34": 2}


def weighted_error_norm(y, y_ref, atol, rtol):
    """sqrt(mean(((y - y_ref)/(|y_ref| rtol + atol))^2)) — the reference's acceptance norm (ode_solver/mod.rs:164-173, threshold 20)."""
    y, y_ref = np.asarray(y, dtype=float), np.asarray(y_ref, dtype=float)
    atol = np.broadcast_to(np.asarray(atol, dtype=float), y_ref.shape[-1:])
    e = (y - y_ref) / (np.abs(y_ref) * rtol + atol)
    return float(np.sqrt(np.mean(e * e)))


def robertson_params(nb, seed=12345):
    """SURVEY §8(d) C2 sweep: k1~logU[0.02,0.08], k2~logU[0.5e4,2e4], k3~logU[1.5e7,6e7]."""
    rng = np.random.default_rng(seed)
    return np.stack([np.exp(rng.uniform(np.log(0.02), np.log(0.08), nb)), np.exp(rng.uniform(np.log(0.5e4), np.log(2e4), nb)),
                     np.exp(rng.uniform(np.log(1.5e7), np.log(6e7), nb))], axis=1)


def times_of(kats, spec_t):
    if isinstance(spec_t, str):
        return [pt["t"] for pt in kats[spec_t]["points"]]
    return list(spec_t)


def heat2d_out(y, mgrid):
    """heat2d.rs:200-205: out = (||u||_2 dx)^2 of a state (last axis = states)"""
    dx = 1.0 / (mgrid - 1.0)
    return (np.linalg.norm(np.asarray(y, dtype=float), axis=-1) * dx) ** 2


def foodweb_out(y, nx):
    """foodweb.rs:699-712: (c1 top-left, c1 bottom-right, c2 top-left, c2 bottom-right) of a state (last axis = states, species interleaved)"""
    y = np.asarray(y, dtype=float)
    br = 2 * (nx - 1) + 2 * nx * (nx - 1)
    return np.stack([y[..., 0], y[..., br], y[..., 1], y[..., br + 1]], axis=-1)
