"""GPU parity tests of the Vector / Matrix trait surface of the HIP backend (rows a2-a4 of SURVEY §8).

Part 1 replays the literal known-answer cases of the reference's backend-generic test generators
(crates/diffsol-la/src/vector/mod.rs:705-1135 `test_batched_*`, matrix/mod.rs) — inputs and expected outputs are the reference's.
Part 2 checks every op on seeded random data against a numpy statement of the same semantics at ragged / non-multiple-of-64 sizes,
bit-exactly (the kernels do no re-association)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import diffsol_amd
    return diffsol_amd


@pytest.fixture(scope="module")
def ctx1(H):
    return H.HipContext(0, nbatch=1)


def V(H, data, ctx):
    return H.HipVec.from_vec(data, ctx)


# ------------------------------------------------------------------ part 1: the reference's literal batched KATs
def test_ref_batched_from_vec_roundtrip_and_bad_length(H, ctx1):  # vector/mod.rs:714-727
    c2 = ctx1.clone_with_nbatch(2)
    v = V(H, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], c2)
    assert len(v) == 3 and v.clone_as_vec().reshape(-1).tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    with pytest.raises(H.DiffsolHipError):
        V(H, [1.0, 2.0, 3.0], c2)


def test_ref_batched_from_element_fill_set_index(H, ctx1):  # :729-735, :791-815
    c3 = ctx1.clone_with_nbatch(3)
    assert H.HipVec.from_element(2, 5.0, c3).clone_as_vec().reshape(-1).tolist() == [5.0] * 6
    v = H.HipVec.zeros(2, c3)
    v.set_index(0, 42.0)
    assert v.clone_as_vec().reshape(-1).tolist() == [42.0, 0.0, 42.0, 0.0, 42.0, 0.0]
    with pytest.raises(H.DiffsolHipError):  # test_batched_get_index_panics
        V(H, [1.0, 2.0, 3.0, 4.0], ctx1.clone_with_nbatch(2)).get_index(0)
    w = H.HipVec.zeros(3, ctx1.clone_with_nbatch(2))
    w.fill(7.0)
    assert w.clone_as_vec().reshape(-1).tolist() == [7.0] * 6


def test_ref_batched_arithmetic(H, ctx1):  # :737-754, :817-824, :876-883, :923-963
    c2 = ctx1.clone_with_nbatch(2)
    y = V(H, [1.0, 2.0, 10.0, 20.0], c2)
    y.axpy(2.0, V(H, [3.0, 4.0, 30.0, 40.0], c2), 1.0)
    assert y.clone_as_vec().reshape(-1).tolist() == [7.0, 10.0, 70.0, 100.0]
    c = V(H, [1.0, 2.0, 3.0, 4.0], c2).add(V(H, [10.0, 20.0, 30.0, 40.0], c2))
    assert c.clone_as_vec().reshape(-1).tolist() == [11.0, 22.0, 33.0, 44.0]
    a = V(H, [2.0, 3.0, 4.0, 5.0], c2)
    a.component_mul_assign(V(H, [10.0, 20.0, 30.0, 40.0], c2))
    assert a.clone_as_vec().reshape(-1).tolist() == [20.0, 60.0, 120.0, 200.0]
    a = V(H, [6.0, 8.0, 12.0, 20.0], c2)
    a.component_div_assign(V(H, [2.0, 4.0, 3.0, 5.0], c2))
    assert a.clone_as_vec().reshape(-1).tolist() == [3.0, 2.0, 4.0, 4.0]
    assert V(H, [10.0, 20.0, 30.0, 40.0], c2).sub(V(H, [1.0, 2.0, 3.0, 4.0], c2)).clone_as_vec().reshape(-1).tolist() == [9.0, 18.0, 27.0, 36.0]
    a = V(H, [10.0, 20.0, 30.0, 40.0], c2)
    a.sub_assign(V(H, [1.0, 2.0, 3.0, 4.0], c2))
    assert a.clone_as_vec().reshape(-1).tolist() == [9.0, 18.0, 27.0, 36.0]
    assert V(H, [1.0, 2.0, 10.0, 20.0], c2).mul(2.0).clone_as_vec().reshape(-1).tolist() == [2.0, 4.0, 20.0, 40.0]
    m = V(H, [1.0, 2.0, 10.0, 20.0], c2)
    m.mul_assign(3.0)
    assert m.clone_as_vec().reshape(-1).tolist() == [3.0, 6.0, 30.0, 60.0]


def test_ref_batched_broadcast(H, ctx1):  # :855-921
    c2 = ctx1.clone_with_nbatch(2)
    y = V(H, [1.0, 2.0, 10.0, 20.0], c2)
    y.axpy(2.0, V(H, [3.0, 4.0], ctx1), 1.0)
    assert y.clone_as_vec().reshape(-1).tolist() == [7.0, 10.0, 16.0, 28.0]
    y = H.HipVec.zeros(2, c2)
    y.copy_from(V(H, [5.0, 7.0], ctx1))
    assert y.clone_as_vec().reshape(-1).tolist() == [5.0, 7.0, 5.0, 7.0]
    a = V(H, [2.0, 3.0, 4.0, 5.0], c2)
    a.component_mul_assign(V(H, [10.0, 20.0], ctx1))
    assert a.clone_as_vec().reshape(-1).tolist() == [20.0, 60.0, 40.0, 100.0]
    a = V(H, [6.0, 8.0, 12.0, 20.0], c2)
    a.component_div_assign(V(H, [2.0, 4.0], ctx1))
    assert a.clone_as_vec().reshape(-1).tolist() == [3.0, 2.0, 6.0, 5.0]
    a = V(H, [1.0, 2.0, 3.0, 4.0], c2)
    a.add_assign(V(H, [10.0, 20.0], ctx1))
    assert a.clone_as_vec().reshape(-1).tolist() == [11.0, 22.0, 13.0, 24.0]
    a = V(H, [10.0, 20.0, 30.0, 40.0], c2)
    a.sub_assign(V(H, [1.0, 2.0], ctx1))
    assert a.clone_as_vec().reshape(-1).tolist() == [9.0, 18.0, 29.0, 38.0]


def test_ref_batched_incompatible_nbatch_is_an_error(H, ctx1):  # :1102-1135 (#[should_panic])
    c2, c3 = ctx1.clone_with_nbatch(2), ctx1.clone_with_nbatch(3)
    a, b = H.HipVec.zeros(2, c2), H.HipVec.zeros(2, c3)
    for op in (lambda: a.axpy(1.0, b, 1.0), lambda: a.copy_from(b), lambda: a.add_assign(b), lambda: a.component_mul_assign(b)):
        with pytest.raises(H.DiffsolHipError):
            op()
    with pytest.raises(H.DiffsolHipError):  # test_batched_axpy_new_bad_length
        a.axpy(1.0, H.HipVec.zeros(3, c2), 1.0)


def test_get_batch_and_write_back(H, ctx1):  # Vector::get_batch / get_batch_mut (vector/mod.rs:227-231; cuda.rs:1285-1308)
    c = ctx1.clone_with_nbatch(5)
    x = np.arange(35, dtype=float).reshape(5, 7)
    v = H.HipVec.from_vec(x, c)
    for b in (0, 3, 4):
        g = v.get_batch(b)
        assert g.nb == 1 and np.array_equal(g.clone_as_vec().reshape(-1), x[b]) and g.get_index(2) == x[b, 2]
    w = H.HipVec.from_vec(-np.arange(7, dtype=float)[None, :], ctx1)
    v.set_batch(2, w)
    x[2] = -np.arange(7)
    assert np.array_equal(v.clone_as_vec(), x)
    with pytest.raises(H.DiffsolHipError):
        v.get_batch(5)


def test_ref_batched_norms(H, ctx1):  # :756-789
    c2 = ctx1.clone_with_nbatch(2)
    assert abs(V(H, [1.0, 0.0, 0.0, 3.0], c2).norm(2) - 3.0) < 1e-12
    assert abs(V(H, [1.0, -2.0, 3.0, 0.0], c2).norm(1) - 3.0) < 1e-12
    x, y = V(H, [1.0, 2.0, 3.0, 4.0], c2), V(H, [1.0, 1.0, 1.0, 1.0], c2)
    atol = V(H, [1e-3, 1e-3], ctx1)
    denom = 1.0 * 1e-2 + 1e-3
    expect = ((3.0 / denom) ** 2 + (4.0 / denom) ** 2) / 2.0
    assert abs(x.squared_norm(y, atol, 1e-2) - expect) < 1e-12 * expect


def test_ref_batched_index_ops(H, ctx1):  # :826-833, :965-1008
    from diffsol_amd.la import HipIndex
    c2 = ctx1.clone_with_nbatch(2)
    v = V(H, [1.0, 2.0, 3.0, 4.0, 5.0, 6.0], c2)
    v.assign_at_indices(HipIndex([0, 2], c2), 0.0)
    assert v.clone_as_vec().reshape(-1).tolist() == [0.0, 2.0, 0.0, 0.0, 5.0, 0.0]
    v1 = H.HipVec.zeros(4, c2)
    v2 = V(H, [10.0, 20.0, 30.0, 40.0, 50.0, 60.0, 70.0, 80.0], c2)
    v1.copy_from_indices(v2, HipIndex([0, 2, 3], c2))
    assert v1.clone_as_vec().reshape(-1).tolist() == [10.0, 0.0, 30.0, 40.0, 50.0, 0.0, 70.0, 80.0]
    r = H.HipVec.zeros(3, c2)
    r.gather(v2, HipIndex([3, 0, 2], c2))
    assert r.clone_as_vec().reshape(-1).tolist() == [40.0, 10.0, 30.0, 80.0, 50.0, 70.0]
    out = H.HipVec.zeros(4, c2)
    V(H, [40.0, 10.0, 30.0, 80.0, 50.0, 70.0], c2).scatter(HipIndex([3, 0, 2], c2), out)
    assert out.clone_as_vec().reshape(-1).tolist() == [10.0, 0.0, 30.0, 40.0, 50.0, 0.0, 70.0, 80.0]


def test_ref_batched_root_finding(H, ctx1):  # :835-853 (+ nalgebra_serial.rs:484-504 single-batch semantics)
    c2 = ctx1.clone_with_nbatch(2)
    found, _, idx = V(H, [1.0, -1.0, 1.0, -1.0], c2).root_finding(V(H, [-1.0, 1.0, -1.0, 1.0], c2))
    assert not found and idx >= 0
    with pytest.raises(H.DiffsolHipError) as e:  # inconsistent batches panic in the reference
        V(H, [1.0, 1.0, 1.0, -1.0], c2).root_finding(V(H, [-1.0, 1.0, 1.0, 1.0], c2))
    assert e.value.code == -5
    found, frac, idx = V(H, [1.0, -2.0, 3.0], ctx1).root_finding(V(H, [0.0, 6.0, -1.0], ctx1))
    assert found and idx == 1 and frac == abs(6.0 / (6.0 + 2.0))


# ------------------------------------------------------------------ part 2: randomised parity against numpy (bit-exact)
SHAPES = [(3, 1), (3, 2), (3, 67), (8, 1000), (1, 4097), (42, 130), (5, 65536 + 3)]


@pytest.mark.parametrize("n,nb", SHAPES)
def test_elementwise_ops_match_numpy_bitwise(H, ctx1, n, nb):
    rng = np.random.default_rng(n * 1000 + nb)
    c = ctx1.clone_with_nbatch(nb)
    a, b = rng.standard_normal((nb, n)), rng.standard_normal((nb, n)) + 3.0
    br = rng.standard_normal((1, n)) + 3.0  # broadcast operand
    A = lambda: V(H, a, c)  # noqa: E731
    B, BR = V(H, b, c), V(H, br, ctx1)
    assert np.array_equal(A().clone_as_vec(), a)
    for name, op, ref in [
        ("add_assign", lambda v: v.add_assign(B), a + b), ("sub_assign", lambda v: v.sub_assign(B), a - b),
        ("mul_assign", lambda v: v.component_mul_assign(B), a * b), ("div_assign", lambda v: v.component_div_assign(B), a / b),
        ("add_assign_bc", lambda v: v.add_assign(BR), a + br), ("div_assign_bc", lambda v: v.component_div_assign(BR), a / br),
        ("axpy", lambda v: v.axpy(0.3, B, -1.7), 0.3 * b + (-1.7) * a), ("axpy_bc", lambda v: v.axpy(0.3, BR, 2.0), 0.3 * br + 2.0 * a),
        ("axpy_beta0", lambda v: v.axpy(0.3, B, 0.0), 0.3 * b), ("scale", lambda v: v.mul_assign(1.0 / 3.0), a * (1.0 / 3.0)),
        ("copy_bc", lambda v: v.copy_from(BR), np.broadcast_to(br, a.shape)), ("fill", lambda v: v.fill(2.5), np.full_like(a, 2.5)),
    ]:
        v = A()
        op(v)
        assert np.array_equal(v.clone_as_vec(), ref), name
    assert np.array_equal(A().add(B).clone_as_vec(), a + b)
    assert np.array_equal(A().sub(BR).clone_as_vec(), a - br)
    alpha = rng.standard_normal(nb)
    v = A()
    v.batched_axpy(alpha, B, 0.5)
    assert np.array_equal(v.clone_as_vec(), alpha[:, None] * b + 0.5 * a)


@pytest.mark.parametrize("n,nb", SHAPES)
def test_reductions_match_numpy(H, ctx1, n, nb):
    rng = np.random.default_rng(7 * n + nb)
    c = ctx1.clone_with_nbatch(nb)
    x, y = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
    atol = np.abs(rng.standard_normal(n)) + 1e-3
    rtol = 1e-2
    def one(b):
        acc = 0.0
        for i in range(n):
            term = x[b, i] / (abs(y[b, i]) * rtol + atol[i])
            acc += term * term
        return acc / n
    per = np.array([one(b) for b in range(min(nb, 300))])
    got, got_per = V(H, x, c).squared_norm(V(H, y, c), V(H, atol, ctx1), rtol, per_batch=True)
    assert np.array_equal(got_per[: len(per)], per)  # same sequential summation order -> bit-exact
    assert got == got_per.max()
    # broadcast y
    got_b = V(H, x, c).squared_norm(V(H, y[:1], ctx1), V(H, atol, ctx1), rtol)
    ref_b = max(np.mean((x[b] / (np.abs(y[0]) * rtol + atol)) ** 2) for b in range(nb))
    assert abs(got_b - ref_b) <= 1e-13 * ref_b
    assert abs(V(H, x, c).norm(2) - np.sqrt((x * x).sum(1)).max()) < 1e-12 * n
    assert abs(V(H, x, c).norm(1) - np.abs(x).sum(1).max()) < 1e-12 * n
    # NaN in one lane propagates (deliberate deviation from the CUDA host-side `>` max, matches the CPU path)
    xn = x.copy()
    xn[nb // 2, 0] = np.nan
    assert np.isnan(V(H, xn, c).squared_norm(V(H, y, c), V(H, atol, ctx1), rtol))


def test_empty_vectors(H, ctx1):  # nstates == 0 (vector/cuda.rs:1365-1367)
    c = ctx1.clone_with_nbatch(4)
    z = H.HipVec.zeros(0, c)
    z.fill(1.0)
    z.add_assign(H.HipVec.zeros(0, c))
    assert z.squared_norm(H.HipVec.zeros(0, c), H.HipVec.zeros(0, ctx1), 1e-3) == 0.0
    assert z.clone_as_vec().shape == (4, 0)


@pytest.mark.parametrize("nb", [1, 3, 200])
def test_matrix_ops_match_numpy(H, ctx1, nb):
    rng = np.random.default_rng(nb)
    c = ctx1.clone_with_nbatch(nb)
    n = 4
    x, y = rng.standard_normal((nb, n, n)), rng.standard_normal((nb, n, n))
    X, Y = H.HipMat.from_array(x, c), H.HipMat.from_array(y, c)
    assert np.array_equal(X.to_array(), x)
    S = H.HipMat.zeros(n, n, c)
    S.scale_add_and_assign(X, -0.37, Y)  # the M - cJ assembly: self = x + beta*y
    assert np.array_equal(S.to_array(), y * (-0.37) + x)
    d = rng.standard_normal((nb, n))
    D = H.HipMat.from_diagonal(V(H, d, c))
    assert np.array_equal(D.to_array(), np.stack([np.diag(d[b]) for b in range(nb)]))
    assert np.array_equal(D.diagonal().clone_as_vec(), d)
    Dbc = H.HipMat.from_diagonal(H.HipVec.from_element(n, 1.0, c))
    assert np.array_equal(Dbc.to_array(), np.broadcast_to(np.eye(n), (nb, n, n)))
    col = rng.standard_normal((nb, n))
    X.set_column(2, V(H, col, c))
    x[:, :, 2] = col
    assert np.array_equal(X.to_array(), x)
    assert np.array_equal(X.column(1).clone_as_vec(), x[:, :, 1])
    X.column_axpy(0.5, 3, 0)
    x[:, :, 0] = x[:, :, 0] + 0.5 * x[:, :, 3]
    assert np.array_equal(X.to_array(), x)
    with pytest.raises(H.DiffsolHipError):
        X.column_axpy(1.0, 1, 1)
    # gemv with nalgebra's accumulation order
    v, w = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
    W = V(H, w, c)
    X.gemv(1.5, V(H, v, c), -0.5, W)
    ref = 1.5 * x[:, :, 0] * v[:, 0:1] + (-0.5) * w
    for j in range(1, n):
        ref = 1.5 * x[:, :, j] * v[:, j:j + 1] + ref
    assert np.array_equal(W.clone_as_vec(), ref)
    # gemm with a broadcast right operand (D[:,0..k+1] * RU, bdf.rs:568-577)
    ru = rng.standard_normal((1, n, n))
    Cm = H.HipMat.zeros(n, n, c)
    Cm.gemm(1.0, X, H.HipMat.from_array(ru, ctx1), 0.0)
    ref = np.empty_like(x)
    for j in range(n):
        acc = 1.0 * x[:, :, 0] * ru[0, 0, j]
        for k in range(1, n):
            acc = 1.0 * x[:, :, k] * ru[0, k, j] + acc
        ref[:, :, j] = acc
    assert np.array_equal(Cm.to_array(), ref)
    assert np.allclose(ref, x @ ru[0], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n,nb", [(128, 8), (130, 3), (512, 37), (1000, 9), (161, 64), (256, 5), (513, 12), (1537, 3)])
def test_norm_of_long_vectors_in_small_ensembles_keeps_the_sequential_summation_bits(H, ctx1, n, nb):
    """n >= 128 and at most 16 384 members run k_squared_norm_wide (8 members per wavefront, the terms of 32 components computed on all 64 lanes, the additions in
    index order on one row group), from 128 components on k_squared_norm_wide (k_squared_norm_team, the workgroup form, is the default only with the fused Newton update; the subprocess test below forces it): every member's value bit for bit the sequential sum, for member counts and lengths that are not multiples of the tile; all four
    broadcast combinations of y and atol; per-member atol; NaN propagation."""
    rng = np.random.default_rng(n + nb)
    c = ctx1.clone_with_nbatch(nb)
    x, y = rng.standard_normal((nb, n)), rng.standard_normal((nb, n))
    atol1, atolb = np.abs(rng.standard_normal(n)) + 1e-3, np.abs(rng.standard_normal((nb, n))) + 1e-3
    rtol = 1e-3

    def ref(yy, aa):
        out = np.empty(nb)
        for b in range(nb):
            yb = yy[b] if yy.shape[0] > 1 else yy[0]
            ab = aa[b] if aa.ndim > 1 else aa
            acc = 0.0
            for i in range(n):
                term = x[b, i] / (abs(yb[i]) * rtol + ab[i])
                acc += term * term
            out[b] = acc / n
        return out

    for yy, yc in ((y, c), (y[:1], ctx1)):
        for aa, ac in ((atol1, ctx1), (atolb, c)):
            got, per = V(H, x, c).squared_norm(V(H, yy, yc), V(H, aa, ac), rtol, per_batch=True)
            want = ref(yy, aa)
            assert np.array_equal(per, want) and got == want.max()
    xn = x.copy()
    xn[nb // 2, n - 1] = np.nan
    assert np.isnan(V(H, xn, c).squared_norm(V(H, y, c), V(H, atol1, ctx1), rtol))


_NORM_TEAM_SCRIPT = """
import ctypes as C, sys
import numpy as np
sys.path.insert(0, {root!r})
import diffsol_amd as H
n, nb = 700, 21
rng = np.random.default_rng(3)
c = H.HipContext(nbatch=nb)
x = rng.standard_normal((nb, n)); y = rng.standard_normal((nb, n)); a = np.abs(rng.standard_normal(n)) * 1e-3 + 1e-6
X, Y, A = H.HipVec.from_vec(x, c), H.HipVec.from_vec(y, c), H.HipVec.from_vec(a[None, :], c.clone_with_nbatch(1))
got = X.squared_norm(Y, A, 1e-4)
t = x / (np.abs(y) * 1e-4 + a[None, :])
ref = (np.cumsum(t * t, axis=1)[:, -1] / n).max()
assert got == ref, (got, ref)
# the SUB form (Newton update + norm in one pass): the staged SDIRK Newton iteration of heat1d with the banded solve's own epilogue switched off
n2 = 512
L = c._L
band = np.zeros((nb, n2, n2)); i = np.arange(n2)
band[:, i, i] = 2.0 + rng.random((nb, n2)); band[:, i[:-1], i[:-1] + 1] = -rng.random((nb, n2 - 1)); band[:, i[1:], i[1:] - 1] = -rng.random((nb, n2 - 1))
lu = H.HipLU(c, n2); lu.factor(H.HipMat.from_array(band, c))
k0 = rng.standard_normal((nb, n2)) * 1e-3
K, PHI, P, YE = H.HipVec.from_vec(k0, c), H.HipVec.from_vec(rng.standard_normal((nb, n2)), c), H.HipVec.from_vec(rng.uniform(0.5, 2.0, (nb, 1)), c), H.HipVec.from_vec(rng.standard_normal((nb, n2)), c)
A2 = H.HipVec.from_vec(np.full((1, n2), 1e-6), c.clone_with_nbatch(1))
out = (C.c_double * 3)()
assert L.dsh_sdirk_newton_iter(c._h, 7, n2, nb, 0.0, 1e-3, 2e-4, K.ptr, K.ptr, PHI.ptr, P.ptr, lu._h, YE.ptr, A2.ptr, 1, 1e-6, out) == 0
np.save(sys.argv[1], np.concatenate([np.asarray(K.clone_as_vec()).ravel(), [out[0]]]))
"""


def test_workgroup_form_of_the_norm_plain_and_with_the_newton_update_gives_the_sequential_sums_bits(tmp_path):
    """k_squared_norm_team without the fused update is never the default (DSH_NORM_TEAM is read once per process), so it ran in no test.  A subprocess forces
    it: the plain norm against numpy's in-order sum, and the SUB form (update + norm, reached with the banded solve's epilogue off) against the default path's iterate
    and norm — bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "norm_team.py"
    script.write_text(_NORM_TEAM_SCRIPT.format(root=root))
    outs = []
    for env_extra in ({"DSH_NORM_TEAM": "1", "DSH_LU_SOLVE_EPI": "0"}, {"DSH_NORM_TEAM": "0", "DSH_LU_SOLVE_EPI": "0"}, {}):
        out = tmp_path / ("o%d.npy" % len(outs))
        r = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_a_context_moves_to_another_host_thread(H):
    """The threading contract of include/diffsol_hip.h (what the Rust shim's `unsafe impl Send` rests on): a context and its objects may move between host threads;
    every entry point re-binds the HIP device of the calling thread itself (no explicit dsh_ctx_bind_thread).  A vector made on the main thread is used, and its
    norm read, on a worker thread, then again on the main thread."""
    import threading
    c = H.HipContext(nbatch=5)
    x = H.HipVec.from_vec(np.arange(15.0).reshape(5, 3), c)
    got = {}

    def worker():
        x.axpy(2.0, x, 1.0)  # x <- 3 x
        got["norm"] = x.norm(1)
        got["vals"] = np.asarray(x.clone_as_vec()).reshape(5, 3)

    t = threading.Thread(target=worker)
    t.start(); t.join()
    assert np.array_equal(got["vals"], 3.0 * np.arange(15.0).reshape(5, 3)) and got["norm"] == (3.0 * np.arange(15.0).reshape(5, 3)).sum(axis=1).max()
    x.axpy(1.0, x, 1.0)
    assert np.array_equal(np.asarray(x.clone_as_vec()).reshape(5, 3), 6.0 * np.arange(15.0).reshape(5, 3))
    assert c._L.dsh_ctx_bind_thread(c._h) == 0  # still exported for callers with their own HIP calls


def test_two_host_threads_share_one_context(H):
    """HipContext / HipVec are Clone + Send in the Rust shim, so safe code can use clones of one context from two threads at once.  The C library
    serialises the calls (DSH_ENTER: the context's lock around every entry point): allocation cache, record ring and scratch stay consistent.  Two threads
    allocate, reduce and free on ONE context concurrently (ctypes releases the GIL inside the calls); every reduction must be its own thread's value."""
    import threading
    c = H.HipContext(nbatch=7)
    errors = []

    def worker(k):
        try:
            base = np.arange(21.0).reshape(7, 3) + 100.0 * k
            for it in range(200):
                x = H.HipVec.from_vec(base, c)            # dsh_malloc (allocation cache) + upload
                y = x.clone()
                y.axpy(1.0, x, 1.0)                       # y = 2 x
                nrm = y.norm(1)                           # reducing launch: record ring + host reduction into the context
                if nrm != (2.0 * base).sum(axis=1).max():
                    errors.append((k, it, nrm))
                    return
                del x, y                                  # dsh_free parks the blocks
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:3]
