"""Transcribe the known-answer DATA the reference's own tests hold for the hot path into tests/golden/reference_kats.json.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_reference_kats.py

What is extracted (data only — numbers out of literal tables / insta snapshots, never source text):
  * solution tables of the Robertson DAE and ODE test models (SUNDIALS IDA / CVODE output quoted in-tree)
      crates/diffsol/src/ode_equations/test_models/robertson.rs:119-133, robertson_ode.rs:107-121
  * the integer work counters pinned by insta snapshots in crates/diffsol/src/ode_solver/bdf.rs and sdirk.rs
      (OdeSolverStatistics + rhs OpStatistics after `test_ode_solver(...)`)
  * the BdfCallable / SdirkCallable unit KATs (crates/diffsol/src/op/bdf.rs:318-361, op/sdirk.rs:338-389)
  * BDF method constants (kappa table bdf.rs:253-260) and the tableaus' defining constants (tableau.rs:41-159)
  * round 6: the 2-D PDE test models of the banded-solver row — solution tables of heat2d (test_models/heat2d.rs:274-287: the model's out
      (||u||_2 dx)^2 at 12 times) and foodweb (test_models/foodweb.rs:996-1050: the corner values of both species at 7 times), and the solver counters of
      test_bdf_faer_sparse_heat2d / test_bdf_faer_sparse_foodweb (bdf.rs:2424-2490)
"""
import json
import os
import re

REF = "/root/reference/crates/diffsol/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def read(path):
    with open(os.path.join(REF, path)) as f:
        return f.read()


def parse_table(src, start_marker):
    """(vec![a, b, c], t) rows following `start_marker`."""
    i = src.index(start_marker)
    block = src[i:src.index("];", i)]
    rows = []
    for m in re.finditer(r"\(vec!\[([^\]]+)\],\s*([-+0-9.eE]+)\)", block):
        vals = [float(v) for v in m.group(1).split(",")]
        rows.append({"t": float(m.group(2)), "y": vals})
    return rows


def parse_table_multiline(src, start_marker):
    """the same for tables whose rows rustfmt broke over several lines: ( vec![ a, b, ], t, )"""
    i = src.index(start_marker)
    block = src[i:src.index("];", i)]
    rows = []
    for m in re.finditer(r"\(\s*vec!\[([^\]]+)\],\s*([-+0-9.eE]+),?\s*\)", block):
        vals = [float(v) for v in m.group(1).split(",") if v.strip()]
        rows.append({"t": float(m.group(2)), "y": vals})
    return rows


def parse_snapshots(src, wanted):
    """For each test fn name in `wanted`: the first two insta yaml blocks after `fn name(`."""
    out = {}
    for name in wanted:
        i = src.index(f"fn {name}(")
        j = src.index("#[test]", i) if "#[test]" in src[i:] else len(src)
        body = src[i:j]
        counters = {}
        for key in ["number_of_linear_solver_setups", "number_of_steps", "number_of_error_test_failures", "number_of_nonlinear_solver_iterations",
                    "number_of_nonlinear_solver_fails", "number_of_linear_solver_setups_from_checkpoint",
                    "number_of_linear_solver_setups_from_first_convergence_fail", "number_of_linear_solver_setups_from_second_convergence_fail",
                    "number_of_linear_solver_setups_from_error_test_fail", "number_of_linear_solver_setups_from_step_success", "number_of_calls",
                    "number_of_jac_muls", "number_of_matrix_evals"]:
            m = re.search(rf"^\s*{key}: (\d+)\s*$", body, re.M)
            if m:
                counters[key] = int(m.group(1))
        out[name] = counters
    return out


def main():
    kats = {"_generated_by": "tests/golden/make_reference_kats.py", "_source": "martinjrobins/diffsol workspace v0.16.2 (/root/reference)"}

    kats["robertson_dae_table"] = {
        "source": "test_models/robertson.rs:119-133", "p": [0.04, 1.0e4, 3.0e7], "rtol": 1e-4, "atol": [1.0e-8, 1.0e-6, 1.0e-6],
        "points": parse_table(read("ode_equations/test_models/robertson.rs"), "fn soln<V: Vector>"),
    }
    kats["robertson_ode_table"] = {
        "source": "test_models/robertson_ode.rs:107-121", "p": [0.04, 1.0e4, 3.0e7], "rtol": 1e-4, "atol": [1.0e-8, 1.0e-14, 1.0e-6],
        "points": parse_table(read("ode_equations/test_models/robertson_ode.rs"), "let mut soln = OdeSolverSolution::default();"),
    }
    assert len(kats["robertson_dae_table"]["points"]) == 13 and len(kats["robertson_ode_table"]["points"]) == 13

    bdf = read("ode_solver/bdf.rs")
    sdirk = read("ode_solver/sdirk.rs")
    kats["bdf_snapshots"] = parse_snapshots(bdf, [
        "bdf_test_nalgebra_exponential_decay", "test_bdf_nalgebra_exponential_decay_algebraic", "test_bdf_nalgebra_robertson",
        "test_bdf_nalgebra_robertson_ode", "test_bdf_nalgebra_dydt_y2", "test_bdf_nalgebra_gaussian_decay"])
    kats["sdirk_snapshots"] = parse_snapshots(sdirk, [
        "test_tr_bdf2_nalgebra_exponential_decay2", "test_esdirk34_nalgebra_exponential_decay", "test_esdirk34_nalgebra_exponential_decay_algebraic",
        "test_tr_bdf2_nalgebra_robertson", "test_esdirk34_nalgebra_robertson", "test_tr_bdf2_nalgebra_robertson_ode"])
    for group in ("bdf_snapshots", "sdirk_snapshots"):
        for name, c in kats[group].items():
            assert len(c) == 13, (name, c)

    # round 6: the reference's 2-D PDE test models (banded Jacobians, half-bandwidth 10 and 20).  Their snapshots are taken with FaerSparseLU and a coloured sparse
    # Jacobian: the ten OdeSolverStatistics counters do not depend on that (up to rounding in the linear solves), number_of_jac_muls does — only the solver counters
    # and the rhs call / matrix evaluation counts are kept for the banded (dense-container) path
    kats["pde2d_snapshots"] = parse_snapshots(bdf, ["test_bdf_faer_sparse_heat2d", "test_bdf_faer_sparse_foodweb"])
    assert len(kats["pde2d_snapshots"]["test_bdf_faer_sparse_heat2d"]) == 13 and len(kats["pde2d_snapshots"]["test_bdf_faer_sparse_foodweb"]) == 10
    kats["heat2d_table"] = {"source": "test_models/heat2d.rs:267-287", "what": "out = (||u||_2 dx)^2, dx = 1/(mgrid-1)", "mgrid": 10, "problem_rtol": 1e-7, "problem_atol": [1e-7],
                            "rtol": 1e-5, "atol": [1e-5], "points": parse_table(read("ode_equations/test_models/heat2d.rs"), "let data = vec![")}
    kats["foodweb_table"] = {"source": "test_models/foodweb.rs:988-1050", "what": "out = (c1 top-left, c1 bottom-right, c2 top-left, c2 bottom-right)", "nx": 10,
                             "problem_rtol": 1e-5, "problem_atol": [1e-5], "h0": 1.0, "rtol": 1e-4, "atol": [1e-4] * 4,
                             "points": parse_table_multiline(read("ode_equations/test_models/foodweb.rs"), "let data = vec![")}
    assert len(kats["heat2d_table"]["points"]) == 12 and len(kats["foodweb_table"]["points"]) == 7 and all(len(r["y"]) == 4 for r in kats["foodweb_table"]["points"])

    # test-problem definitions that go with the snapshots (arguments of the reference's problem constructors)
    kats["snapshot_problems"] = {
        "bdf_test_nalgebra_exponential_decay": {"model": "exponential_decay", "p": [0.1, 1.0], "h0": 1.0, "rtol": 1e-6, "atol": [1e-6], "t": [float(i) for i in range(10)], "method": "bdf"},
        "test_bdf_nalgebra_exponential_decay_algebraic": {"model": "exponential_decay_with_algebraic", "p": [0.1], "h0": 1.0, "rtol": 1e-6, "atol": [1e-6], "t": [i / 10.0 for i in range(10)], "method": "bdf"},
        "test_bdf_nalgebra_robertson": {"model": "robertson", "p": [0.04, 1e4, 3e7], "h0": 1.0, "rtol": 1e-4, "atol": [1e-8, 1e-6, 1e-6], "t": "robertson_dae_table", "method": "bdf"},
        "test_bdf_nalgebra_robertson_ode": {"model": "robertson_ode", "size": 3, "p": [0.04, 1e4, 3e7], "h0": 1.0, "rtol": 1e-4, "atol": [1e-8, 1e-14, 1e-6], "t": "robertson_ode_table", "method": "bdf"},
        "test_bdf_nalgebra_dydt_y2": {"model": "dydt_y2", "size": 10, "p": [], "h0": 1.0, "rtol": 1e-4, "atol": [1e-6], "t": [2.0 * i for i in range(11)], "method": "bdf"},
        "test_bdf_nalgebra_gaussian_decay": {"model": "gaussian_decay", "size": 10, "p": [0.1] * 10, "h0": 1.0, "rtol": 1e-6, "atol": [1e-6], "t": [float(i) for i in range(10)], "method": "bdf"},
        "test_tr_bdf2_nalgebra_exponential_decay2": {"model": "exponential_decay", "p": [0.1, 1.0], "h0": 1.0, "rtol": 1e-6, "atol": [1e-6], "t": [float(i) for i in range(10)], "method": "tr_bdf2"},
        "test_esdirk34_nalgebra_exponential_decay": {"model": "exponential_decay", "p": [0.1, 1.0], "h0": 1.0, "rtol": 1e-6, "atol": [1e-6], "t": [float(i) for i in range(10)], "method": "esdirk34"},
        "test_esdirk34_nalgebra_exponential_decay_algebraic": {"model": "exponential_decay_with_algebraic", "p": [0.1], "h0": 1.0, "rtol": 1e-6, "atol": [1e-6], "t": [i / 10.0 for i in range(10)], "method": "esdirk34"},
        "test_tr_bdf2_nalgebra_robertson": {"model": "robertson", "p": [0.04, 1e4, 3e7], "h0": 1.0, "rtol": 1e-4, "atol": [1e-8, 1e-6, 1e-6], "t": "robertson_dae_table", "method": "tr_bdf2"},
        "test_esdirk34_nalgebra_robertson": {"model": "robertson", "p": [0.04, 1e4, 3e7], "h0": 1.0, "rtol": 1e-4, "atol": [1e-8, 1e-6, 1e-6], "t": "robertson_dae_table", "method": "esdirk34"},
        "test_tr_bdf2_nalgebra_robertson_ode": {"model": "robertson_ode", "size": 1, "p": [0.04, 1e4, 3e7], "h0": 1.0, "rtol": 1e-4, "atol": [1e-8, 1e-14, 1e-6], "t": "robertson_ode_table", "method": "tr_bdf2"},
    }

    # unit KATs of the residual operators (values quoted in the reference tests)
    kats["bdf_callable_kat"] = {"source": "op/bdf.rs:318-361", "model": "exponential_decay", "p": [0.1, 1.0], "c": 0.1, "psi_neg_y0": [1.1, 1.2], "y": [1.0, 1.0],
                                "t": 0.0, "F": [2.11, 2.21], "v": [1.0, 1.0], "Jv": [1.01, 1.01], "J": [[1.01, 0.0], [0.0, 1.01]], "tol": 1e-10}
    kats["sdirk_callable_kat"] = {"source": "op/sdirk.rs:338-389", "model": "exponential_decay", "p": [0.1, 1.0], "c": 0.1, "h": 1.0, "phi": [1.1, 1.2], "y": [1.0, 1.0],
                                  "t": 0.0, "F": [1.12, 1.13], "v": [1.0, 1.0], "Jv": [1.01, 1.01], "J": [[1.01, 0.0], [0.0, 1.01]], "tol": 1e-10}
    kats["sdirk_robertson_jacobian_kat"] = {"source": "op/sdirk.rs:316-336", "model": "robertson", "p": [0.04, 1e4, 3e7], "c": 0.1, "h": 1.3, "phi": [1.1, 1.2, 1.3],
                                            "y": [1.1, 1.2, 1.3], "v": [2.0, 3.0, 4.0], "t": 0.9, "tol": 1e-10}

    # method constants
    m = re.search(r"let kappa: \[Eqn::T; 6\] = \[(.*?)\];", bdf, re.S)
    assert m and "-0.1850" in m.group(1) and "-0.0823" in m.group(1) and "-0.0415" in m.group(1)
    kats["bdf_kappa"] = [0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0]
    tab = read("ode_solver/tableau.rs")
    nums = [float(x.replace("_", "")) for x in re.findall(r"from_f64\((-?0\.[0-9_]+)\)", tab[tab.index("pub fn esdirk34"):tab.index("pub fn tsit45")])]
    kats["esdirk34_constants"] = nums
    assert abs(nums[0] - 0.435866521508459) < 1e-15

    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
