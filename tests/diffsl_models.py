"""DiffSL texts used by the DiffSL tests, and the glue that gives the CPU oracle the same user model the GPU gets.

The texts are written here (the reference's own DiffSL sources are not copied): the expressions follow the arithmetic order of the built-in registry
models (csrc/dsh_models.hpp, oracle/oracle_models.hpp), which in turn follow the reference closures, so `DiffSL model == built-in model` can be
asserted bit for bit wherever the front end adds no arithmetic of its own."""
import hashlib
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ROBERTSON_ODE = """
in = [k1, k2, k3]
k1 { 0.04 } k2 { 10000 } k3 { 30000000 }
u_i { x = 1, y = 0, z = 0 }
F_i {
  -k1 * x + k2 * y * z,
  k1 * x - k2 * y * z - k3 * y * y,
  k3 * y * y,
}
"""

ROBERTSON_DAE = """
in = [k1, k2, k3]
k1 { 0.04 } k2 { 10000 } k3 { 30000000 }
u_i { x = 1, y = 0, z = 0 }
dudt_i { dxdt = 0, dydt = 0, dzdt = 0 }
M_i { dxdt, dydt, 0 }
F_i {
  -k1 * x + k2 * y * z,
  k1 * x - k2 * y * z - k3 * y * y,
  x + y + z - 1,
}
out_i { x, y, z, x + y + z }
"""

# the electrical-circuits primer model with per-member inputs and a threshold event (BASELINE config 5)
RLC = """
in = [R, L, C, V0, omega, ithresh]
R { 100.0 } L { 1.0 } C { 0.001 } V0 { 10 } omega { 100.0 } ithresh { 0.05 }
Vs { V0 * sin(omega * t) }
u_i { iR = 0, iL = 0, iC = 0, V = 0 }
dudt_i { diRdt = 0, diLdt = 0, diCdt = 0, dVdt = 0 }
M_i { 0, diLdt, 0, dVdt }
F_i {
  V - R * iR,
  (Vs - V) / L,
  iL - iR - iC,
  iC / C,
}
stop_i { iR - ithresh }
out_i { V, iR }
"""

LOGISTIC = """
in = [r, k]
r { 1 } k { 1 }
u_i { y = 0.1 }
F_i { r * y * (1 - y / k) }
stop_i { y - 0.9 * k }
"""

# every function of the language once, smooth at the evaluation points used by the derivative check
ZOO = """
in = [a, b]
a { 0.7 } b { 1.3 }
u_i { x = 0.4, y = 0.9, z = 1.7 }
F_i {
  sin(a * x) * cos(y) + tan(0.3 * z) - exp(-x * y) + log(z + b) + log10(y + 2),
  sqrt(x + y * y) * abs(x - z) + sigmoid(a * y) + tanh(x * z) + sinh(0.5 * y) - cosh(0.3 * x),
  arcsinh(x * y) + arccosh(z + 1) + pow(y, b) + pow(x + 2, 3) + min(x * x, y) * max(z, a * x) + copysign(y, -z) + heaviside(x) * z / (b + t),
}
"""


def heat1d(n):
    """test_models/heat1d.rs as DiffSL: D (A u) / h^2, A = tridiag(1, -2, 1), h = 1/(n+1), triangle initial condition."""
    ic = []
    for i in range(n):
        xx = (i + 1) * (1.0 / (n + 1))
        ic.append(repr(2.0 * xx if xx < 0.5 else 2.0 * (1.0 - xx)))
    init = ",\n".join(f"  ({i}): {v}" for i, v in enumerate(ic))
    return f"""
in = [D]
D {{ 1.0 }}
h {{ 1.0 / {float(n + 1)!r} }}
A_ij {{
  (1..{n}, 0..{n - 1}): 1.0,
  (0..{n}, 0..{n}): -2.0,
  (0..{n - 1}, 1..{n}): 1.0,
}}
u_i {{
{init}
}}
heat_i {{ A_ij * u_j }}
F_i {{ D * heat_i / (h * h) }}
"""


def spm(m=20, voltage=False, no_stops=False):
    """no_stops=True: the same equations without stop conditions (forward sensitivities in the device-resident kernels need a model without root functions).
    voltage=True: the stop conditions of the battery primer (terminal voltage leaves [3.105, 4.1] V) instead of the cheap surface-concentration limits.
    The single-particle model of the battery primer (n = 2 + 2m) written as DiffSL from its formulas: spherical finite-volume Laplacians as sparse
    matrices, flux terms on the outer shells, terminal-voltage stop conditions.  Constants as in the built-in model (oracle_models.hpp Spm)."""
    def lap(scale):
        rows = []
        dr = 1.0 / m
        for k in range(m):
            i0, i1 = float(k), float(k + 1)
            vol = i1 * i1 * i1 - i0 * i0 * i0
            lower = 3.0 * i0 * i0 / vol / (dr * dr) * scale
            upper = 3.0 * i1 * i1 / vol / (dr * dr) * scale if k + 1 < m else 0.0
            if k > 0:
                rows.append(f"  ({k},{k - 1}): {lower!r}")
            rows.append(f"  ({k},{k}): {-(lower + upper)!r}")
            if k + 1 < m:
                rows.append(f"  ({k},{k + 1}): {upper!r}")
        return ",\n".join(rows)
    a, b = 2 + m, 2 + 2 * m
    surf = lambda name: f"{name}_ij {{ ({0},{m - 2}): -0.5, ({0},{m - 1}): 1.5 }}"
    ocp_p = ("2.16216 + 0.07645 * tanh(30.834 - 57.858397200000006 * sp) + 2.1581 * tanh(52.294 - 53.412228 * sp) - 0.14169 * tanh(11.0923 - 21.0852666 * sp) + "
             "0.2051 * tanh(1.4684 - 5.829105600000001 * sp) + 0.2531 * tanh(4.291641337386018 - 8.069908814589667 * sp) - 0.02167 * tanh(-87.5 + 177.0 * sp)")
    if voltage:  # the terminal voltage as the built-in model writes it (oracle_models.hpp spm_voltage): Butler-Volmer overpotentials + open-circuit fits
        clamp = lambda v, lo, hi: f"max(min({v}, {hi!r}), {lo!r})"
        ocp_n = ("0.194 + 1.5 * exp(-120.0 * stn) + 0.0351 * tanh(-3.44578313253012 + 12.048192771084336 * stn) - 0.0045 * tanh(-7.1344537815126055 + 8.403361344537815 * stn) - "
                 "0.035 * tanh(-18.466 + 20.0 * stn) - 0.0147 * tanh(-14.705882352941176 + 29.41176470588235 * stn) - 0.102 * tanh(-1.3661971830985917 + 7.042253521126761 * stn) - "
                 "0.022 * tanh(-54.8780487804878 + 60.975609756097555 * stn) - 0.011 * tanh(-5.486725663716814 + 44.24778761061947 * stn) + "
                 "0.0155 * tanh(-3.6206896551724133 + 34.48275862068965 * stn) + 0.000001 * (1.0 / stn + 1.0 / (-1.0 + stn))")
        ocp_pv = ocp_p.replace("sp", "stp") + " + 0.000001 * (1.0 / stp + 1.0 / (-1.0 + stp))"
        stops = f"""
SP_ij {{ (0,{m - 2}): -0.4999999999999983, (0,{m - 1}): 1.4999999999999982 }}
SN_ij {{ (0,{m - 2}): -0.4999999999999983, (0,{m - 1}): 1.4999999999999984 }}
CP_ij {{ (0,{m - 2}): -25608.96286546366, (0,{m - 1}): 76826.88859639116 }}
CN_ij {{ (0,{m - 2}): -12491.630996921805, (0,{m - 1}): 37474.892990765504 }}
spr_i {{ SP_ij * cp_j }}
snr_i {{ SN_ij * cn_j }}
cpr_i {{ CP_ij * cp_j }}
cnr_i {{ CN_ij * cn_j }}
stp {{ {clamp("spr_i", 1e-10, 0.9999999999)} }}
stn {{ {clamp("snr_i", 1e-10, 0.9999999999)} }}
cps {{ {clamp("cpr_i", 0.000512179257309275, 51217.92521874824)} }}
cns {{ {clamp("cnr_i", 0.000249832619938437, 24983.261744011077)} }}
etap {{ 0.05138515824298745 * arcsinh((-2.3508116177110145 * current) / (2.0 * ((1.8973665961010275e-05 * sqrt(cps)) * sqrt(51217.9257309275 - cps)))) }}
etan {{ 0.05138515824298745 * arcsinh((1.9590096814258458 * current) / (2.0 * ((0.0006324555320336759 * sqrt(cns)) * sqrt(24983.2619938437 - cns)))) }}
volt {{ (etap + ({ocp_pv})) - (etan + ({ocp_n})) }}
stop_i {{ -3.105 + volt, 4.1 - volt }}
out_i {{ volt }}
"""
    else:
        stops = f"""
stop_i {{
  sn_i - 0.05,
  sp_i - 0.99,
}}
out_i {{ sn_i, sp_i, {ocp_p.replace("sp", "sp_i")} }}
"""
    if no_stops:
        stops = ""
    return f"""
in = [current]
current {{ 1.0 }}
Aneg_ij {{
{lap(0.39e-3)}
}}
Apos_ij {{
{lap(1.0e-3)}
}}
eneg_i {{ (0:{m - 1}): 0.0, ({m - 1}): {3.2835305549534856e-12 * -520607810.21082705!r} }}
epos_i {{ (0:{m - 1}): 0.0, ({m - 1}): {4.106800547504748e-12 * 243644455.17866704!r} }}
{surf("S")}
u_i {{
  q = 0.0,
  thr = 0.0,
  (2:{a}): cn = 0.8000000000000016,
  ({a}:{b}): cp = 0.6000000000000001,
}}
ln_i {{ Aneg_ij * cn_j }}
lp_i {{ Apos_ij * cp_j }}
sn_i {{ S_ij * cn_j }}
sp_i {{ S_ij * cp_j }}
F_i {{
  0.0002777777777777778 * current,
  0.0002777777777777778 * abs(current),
  ln_i + eneg_i * current,
  lp_i + epos_i * current,
}}
{stops}"""


def spm_dae(m=20):
    """BASELINE configs[3] as it is worded — "physics-based battery SPM DAE (singular mass matrix)": the single-particle model of the battery primer with the
    TERMINAL VOLTAGE as an algebraic state (the way benches/pybamm_dfn.diffsl carries its voltage): n = 2 + 2m + 1, M = diag(1, ..., 1, 0, 1, ..., 1),
    0 = volt(c_neg surface, c_pos surface, I) - V, stop when V leaves [3.105, 4.1].  State order: the two charge counters, the negative particle's shells
    (centre to surface), V, the positive particle's shells SURFACE TO CENTRE — so that V sits between the four shell values it depends on and the
    Jacobian keeps a bandwidth of 2 (the lane-per-member banded kernels).  Same constants and formulas as spm(m, voltage=True); V starts at a guess and is
    made consistent by the solver (StateRefMut::set_consistent)."""
    def lap_rows(scale, reverse):
        rows = []
        dr = 1.0 / m
        for k in range(m):
            i0, i1 = float(k), float(k + 1)
            vol = i1 * i1 * i1 - i0 * i0 * i0
            lower = 3.0 * i0 * i0 / vol / (dr * dr) * scale
            upper = 3.0 * i1 * i1 / vol / (dr * dr) * scale if k + 1 < m else 0.0
            ix = (lambda a: m - 1 - a) if reverse else (lambda a: a)
            if k > 0:
                rows.append(f"  ({ix(k)},{ix(k - 1)}): {lower!r}")
            rows.append(f"  ({ix(k)},{ix(k)}): {-(lower + upper)!r}")
            if k + 1 < m:
                rows.append(f"  ({ix(k)},{ix(k + 1)}): {upper!r}")
        return ",\n".join(rows)
    a = 2 + m
    clamp = lambda v, lo, hi: f"max(min({v}, {hi!r}), {lo!r})"
    ocp_p = ("2.16216 + 0.07645 * tanh(30.834 - 57.858397200000006 * stp) + 2.1581 * tanh(52.294 - 53.412228 * stp) - 0.14169 * tanh(11.0923 - 21.0852666 * stp) + "
             "0.2051 * tanh(1.4684 - 5.829105600000001 * stp) + 0.2531 * tanh(4.291641337386018 - 8.069908814589667 * stp) - 0.02167 * tanh(-87.5 + 177.0 * stp)"
             " + 0.000001 * (1.0 / stp + 1.0 / (-1.0 + stp))")
    ocp_n = ("0.194 + 1.5 * exp(-120.0 * stn) + 0.0351 * tanh(-3.44578313253012 + 12.048192771084336 * stn) - 0.0045 * tanh(-7.1344537815126055 + 8.403361344537815 * stn) - "
             "0.035 * tanh(-18.466 + 20.0 * stn) - 0.0147 * tanh(-14.705882352941176 + 29.41176470588235 * stn) - 0.102 * tanh(-1.3661971830985917 + 7.042253521126761 * stn) - "
             "0.022 * tanh(-54.8780487804878 + 60.975609756097555 * stn) - 0.011 * tanh(-5.486725663716814 + 44.24778761061947 * stn) + "
             "0.0155 * tanh(-3.6206896551724133 + 34.48275862068965 * stn) + 0.000001 * (1.0 / stn + 1.0 / (-1.0 + stn))")
    return f"""
in = [current]
current {{ 1.0 }}
Aneg_ij {{
{lap_rows(0.39e-3, False)}
}}
Aposr_ij {{
{lap_rows(1.0e-3, True)}
}}
eneg_i {{ (0:{m - 1}): 0.0, ({m - 1}): {3.2835305549534856e-12 * -520607810.21082705!r} }}
eposr_i {{ (0): {4.106800547504748e-12 * 243644455.17866704!r}, (1:{m}): 0.0 }}
SP_ij {{ (0,0): 1.4999999999999982, (0,1): -0.4999999999999983 }}
SN_ij {{ (0,{m - 2}): -0.4999999999999983, (0,{m - 1}): 1.4999999999999984 }}
CP_ij {{ (0,0): 76826.88859639116, (0,1): -25608.96286546366 }}
CN_ij {{ (0,{m - 2}): -12491.630996921805, (0,{m - 1}): 37474.892990765504 }}
u_i {{
  q = 0.0,
  thr = 0.0,
  (2:{a}): cn = 0.8000000000000016,
  V = 4.0,
  ({a + 1}:{a + 1 + m}): cpr = 0.6000000000000001,
}}
dudt_i {{
  dq = 0.0,
  dthr = 0.0,
  (2:{a}): dcn = 0.0,
  dV = 0.0,
  ({a + 1}:{a + 1 + m}): dcpr = 0.0,
}}
ln_i {{ Aneg_ij * cn_j }}
lp_i {{ Aposr_ij * cpr_j }}
spr_i {{ SP_ij * cpr_j[0:2] }}
snr_i {{ SN_ij * cn_j }}
cpr2_i {{ CP_ij * cpr_j[0:2] }}
cnr_i {{ CN_ij * cn_j }}
stp {{ {clamp("spr_i", 1e-10, 0.9999999999)} }}
stn {{ {clamp("snr_i", 1e-10, 0.9999999999)} }}
cps {{ {clamp("cpr2_i", 0.000512179257309275, 51217.92521874824)} }}
cns {{ {clamp("cnr_i", 0.000249832619938437, 24983.261744011077)} }}
etap {{ 0.05138515824298745 * arcsinh((-2.3508116177110145 * current) / (2.0 * ((1.8973665961010275e-05 * sqrt(cps)) * sqrt(51217.9257309275 - cps)))) }}
etan {{ 0.05138515824298745 * arcsinh((1.9590096814258458 * current) / (2.0 * ((0.0006324555320336759 * sqrt(cns)) * sqrt(24983.2619938437 - cns)))) }}
volt {{ (etap + ({ocp_p})) - (etan + ({ocp_n})) }}
M_i {{
  dq,
  dthr,
  dcn_i,
  0,
  dcpr_i,
}}
F_i {{
  0.0002777777777777778 * current,
  0.0002777777777777778 * abs(current),
  ln_i + eneg_i * current,
  volt - V,
  lp_i + eposr_i * current,
}}
stop_i {{ V - 3.105, 4.1 - V }}
out_i {{ V }}
"""


def random_expr(rng, depth, names):
    """(DiffSL text, python callable on a dict of values) of a random expression that stays smooth and finite for positive inputs."""
    if depth == 0 or rng.random() < 0.25:
        if rng.random() < 0.3:
            c = float(np.round(rng.uniform(0.2, 3.0), 3))
            return repr(c), (lambda env, c=c: c)
        nm = names[int(rng.integers(len(names)))]
        return nm, (lambda env, nm=nm: env[nm])
    kind = rng.choice(["+", "-", "*", "/", "neg", "f1", "f2"])
    a_s, a_f = random_expr(rng, depth - 1, names)
    if kind in "+-*/":
        b_s, b_f = random_expr(rng, depth - 1, names)
        if kind == "/":  # keep the denominator away from zero
            return f"({a_s} / (1.5 + abs({b_s})))", (lambda env: a_f(env) / (1.5 + abs(b_f(env))))
        op = {"+": np.add, "-": np.subtract, "*": np.multiply}[kind]
        return f"({a_s} {kind} {b_s})", (lambda env: op(a_f(env), b_f(env)))
    if kind == "neg":
        return f"(-{a_s})", (lambda env: -a_f(env))
    if kind == "f1":
        name, fn, wrap = [("sin", np.sin, "{}"), ("cos", np.cos, "{}"), ("tanh", np.tanh, "{}"), ("exp", np.exp, "0.1 * {}"), ("sqrt", np.sqrt, "1.0 + abs({})"),
                          ("log", np.log, "1.0 + abs({})"), ("arcsinh", np.arcsinh, "{}"), ("sigmoid", lambda v: 1.0 / (1.0 + np.exp(-v)), "{}")][int(rng.integers(8))]
        inner = {"{}": a_f, "0.1 * {}": (lambda env: 0.1 * a_f(env)), "1.0 + abs({})": (lambda env: 1.0 + abs(a_f(env)))}[wrap]
        return f"{name}({wrap.format(a_s)})", (lambda env: fn(inner(env)))
    b_s, b_f = random_expr(rng, depth - 1, names)
    name, fn = [("min", min), ("max", max)][int(rng.integers(2))]
    return f"{name}({a_s}, {b_s})", (lambda env: fn(a_f(env), b_f(env)))



_cache = {}


HOST_MODEL_DIR = os.path.join(ROOT, "oracle", "_build", "host_models")  # git-ignored (_build/), travels with the tree like the other built files


def host_model_so(code, opt="-O2", model_index=0):
    """DiffSL text -> CPU model library (product front end, Target::HostC) -> g++ -> oracle/_build/host_models/<sha1>.so, compiled once per generated source
    (__graft_entry__.build() pre-compiles the ones bench.py asks for).  Returns (path, dims)."""
    from diffsol_amd import diffsl
    src, dims, _ = diffsl.generate(code, diffsl.TARGET_HOST_C, model_index)
    flags = [opt, "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"]
    key = hashlib.sha1((src + "#" + " ".join(flags)).encode()).hexdigest()[:20]
    so = os.path.join(HOST_MODEL_DIR, key + ".so")
    if not os.path.exists(so):
        os.makedirs(HOST_MODEL_DIR, exist_ok=True)
        d = tempfile.mkdtemp(prefix="dsl_host_")
        cpp, tmp = os.path.join(d, "model.cpp"), os.path.join(d, "libmodel.so")
        with open(cpp, "w") as f:
            f.write(src)
        subprocess.run(["g++"] + flags + ["-I", os.path.join(ROOT, "include"), "-o", tmp, cpp], check=True)
        import shutil
        shutil.move(tmp, so + ".tmp%d" % os.getpid())
        os.replace(so + ".tmp%d" % os.getpid(), so)  # atomic: several ranks / xdist workers may want the same model
        shutil.rmtree(d, ignore_errors=True)
    return so, dims


def host_model(O, code, opt="-O2", model_index=0):
    """DiffSL text -> CPU model library -> registered with the oracle.  Returns the oracle id."""
    key = hashlib.sha1((code + "#N=%d#%s" % (model_index, opt)).encode()).hexdigest()
    if key in _cache:
        return _cache[key]
    so, dims = host_model_so(code, opt, model_index)
    mid = O.load_external_model(so)
    assert O.model_dims(mid)["n"] == dims["n"]
    _cache[key] = mid
    return mid


# run-time-sized DAE: heat conduction on 10 interior nodes with the two boundary values as algebraic states (singular mass matrix, n = 12)
HEAT_DAE = """
in = [D]
D { 1.0 }
A_ij { (1..10, 0..9): 1.0, (0..10, 0..10): -2.0, (0..9, 1..10): 1.0 }
eleft_i { (0): 1.0, (1:10): 0.0 }
eright_i { (0:9): 0.0, (9): 1.0 }
u_i { a = 0, (1:11): y = 1, b = 0 }
dudt_i { dadt = 0, (1:11): dydt = 0, dbdt = 0 }
lap_i { A_ij * y_j }
M_i { 0, dydt_i, 0 }
F_i { a, D * (lap_i + eleft_i * a + eright_i * b) * 121.0, b }
out_i { y_i * y_i }
"""


def oscillators(m):
    """n = 2 m states: m damped oscillator pairs (x_k, z_k) with frequency w >> damping a, every x coupled to every other through a DENSE m x m block.  The Jacobian has
    no band structure, and w c > 1 + a c makes the partial pivoting of M - c J interchange rows (the off-diagonal w beats the diagonal): what the workgroup-per-member
    integrator (64 < n <= 140) is tested on."""
    return f"""
in = [w, a, eps]
w {{ 50.0 }}
a {{ 1.0 }}
eps {{ 0.01 }}
S_ij {{ (0:{m}, 0:{m}): 1.0 }}
u_i {{ (0:{m}): x = 1.0, ({m}:{2 * m}): z = 0.0 }}
sx_i {{ S_ij * x_j }}
F_i {{ (0:{m}): -w * z_i - a * x_i - eps * sx_i, ({m}:{2 * m}): w * x_i - a * z_i }}
"""


def heat2d(m, scale_input=True):
    """test_models/heat2d.rs:19-100 (heat2d_diffsl_problem) as the reference's test builds it: the closure model's Jacobian, mass matrix and initial state written
    out as sparse DiffSL tensors — D_ij (5-point stencil rows 1/dx^2 (1, 1, -4, 1, 1), identity rows on the boundary), Mass_ij (diagonal 1 / 0), init_i — and
    F_i = D_ij y_j, M_i = Mass_ij dydt_j, out_i = dx^2 y_j y_j.  Entries column by column like `triplet_iter` of the reference's CSC matrix.  scale_input: an input
    `s` (default 1) multiplies F so that ensembles have distinct members (the reference's text has no input)."""
    n = m * m
    dx = 1.0 / (m - 1.0)
    coeff = 1.0 / (dx * dx)
    four = 4.0
    bnd = lambda loc: (loc // m in (0, m - 1)) or (loc % m in (0, m - 1))
    cols = {}
    for loc in range(n):
        if bnd(loc):
            cols.setdefault(loc, []).append((loc, 1.0))
        else:
            for j, v in ((loc - m, coeff), (loc - 1, coeff), (loc, coeff * (0.0 - four)), (loc + 1, coeff), (loc + m, coeff)):
                cols.setdefault(j, []).append((loc, v))
    d_rows = ",\n".join(f"  ({i}, {j}): {v!r}" for j in sorted(cols) for i, v in sorted(cols[j]))
    mass = ",\n".join(f"  ({i}, {i}): {0 if bnd(i) else 1}" for i in range(n))
    init = []
    for loc in range(n):
        jy, ix = divmod(loc, m)
        yfact, xfact = dx * jy, dx * ix
        init.append(0.0 if bnd(loc) else 16.0 * xfact * (1.0 - xfact) * yfact * (1.0 - yfact))
    init_rows = ",\n".join(f"  {v!r}" for v in init)
    head = "in = [s]\ns { 1.0 }\n" if scale_input else ""
    f_expr = "s * (D_ij * y_j)" if scale_input else "D_ij * y_j"
    return f"""
{head}D_ij {{
{d_rows}
}}
Mass_ij {{
{mass}
}}
init_i {{
{init_rows}
}}
u_i {{
  y = init_i,
}}
dudt_i {{
  (0:{n}): dydt = 0,
}}
M_i {{
  Mass_ij * dydt_j,
}}
F_i {{
  {f_expr},
}}
out_i {{
  {dx * dx!r} * y_j * y_j,
}}
"""


def foodweb(nx):
    """test_models/foodweb.rs:26-146 (foodweb_diffsl_problem): the food web as the reference's test writes it in DiffSL — the two species in separate blocks (c1 = prey,
    c2 = predators), the 5-point diffusion operator with Neumann mirror points as a sparse D_ij (the Jacobian of FoodWebDiff, foodweb.rs:900-940), the grid coordinates
    as vectors, reaction terms element-wise with sin / pow, predators algebraic, corner values as out_i."""
    n = nx * nx
    dx = 1.0 / (nx - 1.0)
    cox = coy = 1.0 / (dx * dx)
    rows = {}
    for jy in range(nx):
        idyu = nx if jy != nx - 1 else -nx
        idyl = nx if jy != 0 else -nx
        for jx in range(nx):
            idxu = 1 if jx != nx - 1 else -1
            idxl = 1 if jx != 0 else -1
            loc = jx + nx * jy
            r = rows.setdefault(loc, {})
            for j, v in ((loc + idyu, coy), (loc, -coy), (loc, -coy), (loc - idyl, coy), (loc + idxu, cox), (loc, -cox), (loc, -cox), (loc - idxl, cox)):
                r[j] = r.get(j, 0.0) + v
    cols = {}
    for i, r in rows.items():
        for j, v in r.items():
            cols.setdefault(j, []).append((i, v))
    d_rows = ",\n".join(f"  ({i}, {j}): {v!r}" for j in sorted(cols) for i, v in sorted(cols[j]))
    xx = ",\n".join(f"  {(loc % nx) * dx!r}" for loc in range(n))
    yy = ",\n".join(f"  {(loc // nx) * dx!r}" for loc in range(n))
    return f"""
AA {{ 1.0 }}
EE {{ 10000.0 }}
GG {{ 0.5e-6 }}
BB {{ 1.0 }}
ALPHA {{ 50.0 }}
BETA {{ 1000.0 }}
PI {{ 3.141592653589793 }}
DPREY {{ 1.0 }}
DPRED {{ 0.05 }}
D_ij {{
{d_rows}
}}
xx_i {{
{xx}
}}
yy_i {{
{yy}
}}
tl_i {{
  (0): 1.0,
  (1:{n}): 0.0,
}}
br_i {{
  (0:{n - 1}): 0.0,
  ({n - 1}): 1.0,
}}
b_i {{
  (1.0 + ALPHA * xx_i * yy_i + BETA * sin(4.0 * PI * xx_i) * sin(4.0 * PI * yy_i))
}}
u_i {{
  c1 = 10.0 + pow(16.0 * xx_i * (1.0 - xx_i) * yy_i * (1.0 - yy_i), 2),
  ({n}:{2 * n}): c2 = 1.0e5,
}}
dudt_i {{
  (0:{n}): dc1dt = 0,
  ({n}:{2 * n}): dc2dt = 0,
}}
M_i {{
  dc1dt_i,
  ({n}:{2 * n}): 0,
}}
c1diff_i {{
  DPREY * D_ij * c1_j,
}}
c2diff_i {{
  DPRED * D_ij * c2_j,
}}
F_i {{
  c1diff_i + c1_i * (BB * b_i - AA * c1_i - GG * c2_i),
  c2diff_i + c2_i * (-BB * b_i + EE * c1_i - AA * c2_i),
}}
out_i {{
  tl_j * c1_j,
  br_j * c1_j,
  tl_j * c2_j,
  br_j * c2_j,
}}
"""
