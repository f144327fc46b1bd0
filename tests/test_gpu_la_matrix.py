"""The reference's batched Matrix / DenseMatrix test cases (crates/diffsol-la/src/matrix/mod.rs:730-1975, the `test_batched_*` generators that
`generate_matrix_tests_batched!` / `generate_dense_matrix_tests_batched!` instantiate for the CUDA backend at matrix/cuda.rs:1474-1490), replayed
with their literal inputs and expected outputs against the HIP backend through the C ABI.  `M::from_vec(nrows, ncols, data)` and
`try_from_triplets` with the index order (0,0),(1,0),(0,1),(1,1) both lay a batch member out column-major, batch after batch."""
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


@pytest.fixture(scope="module")
def ctx2(ctx1):
    return ctx1.clone_with_nbatch(2)


def from_vec(H, nr, nc, data, ctx):
    a = np.asarray(data, dtype=float).reshape(ctx.nbatch, nc, nr)  # [b][col][row]
    return H.HipMat.from_array(np.transpose(a, (0, 2, 1)), ctx)


def vals(m):
    """triplet_values order: batch after batch, column-major"""
    return np.transpose(m.to_array(), (0, 2, 1)).reshape(-1).tolist()


def vec(H, data, ctx):
    return H.HipVec.from_vec(np.asarray(data, dtype=float).reshape(ctx.nbatch, -1), ctx)


def flat(v):
    return v.clone_as_vec().reshape(-1).tolist()


def test_batched_zeros_and_from_vec(H, ctx2):  # :730-738, :951-977
    a = H.HipMat.zeros(2, 3, ctx2)
    assert (a.nrows, a.ncols) == (2, 3) and all(v == 0.0 for v in vals(a))
    a = from_vec(H, 2, 2, [1, 3, 2, 4, 5, 7, 6, 8], ctx2)
    assert a.to_array()[0].tolist() == [[1.0, 2.0], [3.0, 4.0]] and a.to_array()[1].tolist() == [[5.0, 6.0], [7.0, 8.0]]


def test_batched_gemv_and_broadcasts(H, ctx1, ctx2):  # :740-808
    a = from_vec(H, 2, 2, [1, 3, 2, 4, 5, 7, 6, 8], ctx2)
    y = H.HipVec.zeros(2, ctx2)
    a.gemv(1.0, vec(H, [1, 2, 1, 1], ctx2), 0.0, y)
    assert flat(y) == [5.0, 11.0, 11.0, 15.0]
    a.gemv(1.0, vec(H, [1, 2], ctx1), 0.0, y)  # x broadcast
    assert flat(y) == [5.0, 11.0, 17.0, 23.0]
    a1 = from_vec(H, 2, 2, [1, 3, 2, 4], ctx1)  # matrix broadcast
    a1.gemv(1.0, vec(H, [1, 2, 3, 4], ctx2), 0.0, y)
    assert flat(y) == [5.0, 11.0, 11.0, 25.0]


def test_batched_from_diagonal(H, ctx2):  # :810-833, :1944-1960
    a = H.HipMat.from_diagonal(vec(H, [2, 3, 4, 5], ctx2))
    assert (a.nrows, a.ncols) == (2, 2)
    assert a.to_array().tolist() == [[[2.0, 0.0], [0.0, 3.0]], [[4.0, 0.0], [0.0, 5.0]]]


def test_batched_copy_from_set_column_scale_add(H, ctx2):  # :835-949
    a = from_vec(H, 2, 2, [1, 2, 3, 4, 5, 6, 7, 8], ctx2)
    b = H.HipMat.zeros(2, 2, ctx2)
    b.copy_from(a)
    assert vals(b) == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
    z = H.HipMat.zeros(2, 2, ctx2)
    z.set_column(0, vec(H, [5, 6, 7, 8], ctx2))
    assert vals(z) == [5.0, 6.0, 0.0, 0.0, 7.0, 8.0, 0.0, 0.0]
    x = from_vec(H, 2, 2, [1, 2, 3, 4, 5, 6, 7, 8], ctx2)
    y = from_vec(H, 2, 2, [10, 20, 30, 40, 50, 60, 70, 80], ctx2)
    r = H.HipMat.zeros(2, 2, ctx2)
    r.copy_from(x)
    r.scale_add_and_assign(x, 2.0, y)
    assert vals(r) == [21.0, 42.0, 63.0, 84.0, 105.0, 126.0, 147.0, 168.0]


def test_batched_gemm_and_broadcasts(H, ctx1, ctx2):  # :979-1021, :1237-1305, :1904-1942
    a = from_vec(H, 2, 2, [1, 0, 0, 1, 2, 0, 0, 2], ctx2)
    b = from_vec(H, 2, 2, [3, 5, 4, 6, 1, 1, 1, 1], ctx2)
    c = H.HipMat.zeros(2, 2, ctx2)
    c.gemm(1.0, a, b, 0.0)
    assert c.to_array().tolist() == [[[3.0, 4.0], [5.0, 6.0]], [[2.0, 2.0], [2.0, 2.0]]]
    a = from_vec(H, 2, 2, [1, 0, 0, 1, 2, 0, 0, 3], ctx2)
    c.gemm(1.0, a, from_vec(H, 2, 2, [1, 3, 2, 4], ctx1), 0.0)  # B broadcast
    assert c.to_array().tolist() == [[[1.0, 2.0], [3.0, 4.0]], [[2.0, 4.0], [9.0, 12.0]]]
    c.gemm(1.0, from_vec(H, 2, 2, [1, 0, 0, 2], ctx1), b, 0.0)  # A broadcast
    assert c.to_array().tolist() == [[[3.0, 4.0], [10.0, 12.0]], [[1.0, 1.0], [2.0, 2.0]]]
    a = from_vec(H, 2, 2, [1, 3, 2, 4, 2, 1, 0, 3], ctx2)
    b = from_vec(H, 2, 2, [2, 1, 0, 3, 1, 0, 2, 1], ctx2)
    assert a.mat_mul(b).to_array()[0].tolist() == [[4.0, 6.0], [10.0, 12.0]]


def test_batched_column_views(H, ctx1, ctx2):  # :1023-1235, :1307-1433
    a = from_vec(H, 2, 3, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], ctx2)
    view = a.columns(0, 2)
    owned = view.into_owned()
    assert (view.nrows, view.ncols, owned.nrows, owned.ncols) == (2, 2, 2, 2)
    y = H.HipVec.zeros(2, ctx2)
    a.columns(0, 2).gemv_o(1.0, vec(H, [1, 1, 1, 1], ctx2), 0.0, y)
    assert flat(y) == [4.0, 6.0, 16.0, 18.0]
    diff = from_vec(H, 2, 3, [1, 4, 2, 5, 3, 6, 7, 10, 8, 11, 9, 12], ctx2)
    diff.columns(0, 2).gemv_o(1.0, vec(H, [1, 1, 2, 2], ctx2), 0.0, y)
    assert flat(y) == [3.0, 9.0, 30.0, 42.0]
    diff.columns(0, 2).gemv_o(1.0, vec(H, [1, 1], ctx1), 0.0, y)  # x broadcast
    assert flat(y) == [3.0, 9.0, 15.0, 21.0]
    d1 = from_vec(H, 2, 3, [1, 4, 2, 5, 3, 6], ctx1)  # matrix view with nbatch 1 broadcasts to x / y with nbatch 2
    for method in ("gemv_v", "gemv_o"):
        getattr(d1.columns(0, 2), method)(1.0, vec(H, [1, 1, 2, 2], ctx2), 0.0, y)
        assert flat(y) == [3.0, 9.0, 6.0, 18.0]
    r = from_vec(H, 2, 2, [1, 0, 0, 1, 2, 0, 0, 2], ctx2)
    result = H.HipMat.zeros(2, 3, ctx2)
    result.columns(0, 2).gemm_vo(1.0, diff.columns(0, 2), r, 0.0)
    assert result.to_array()[:, :, :2].tolist() == [[[1.0, 2.0], [4.0, 5.0]], [[14.0, 16.0], [20.0, 22.0]]] and (result.to_array()[:, :, 2] == 0).all()
    result = H.HipMat.zeros(2, 3, ctx2)
    result.columns(0, 2).gemm_vo(1.0, diff.columns(0, 2), from_vec(H, 2, 2, [1, 0, 0, 1], ctx1), 0.0)  # R broadcast
    assert result.to_array()[:, :, :2].tolist() == [[[1.0, 2.0], [4.0, 5.0]], [[7.0, 8.0], [10.0, 11.0]]]
    result = H.HipMat.zeros(2, 3, ctx2)
    result.columns(0, 2).gemm_vo(1.0, d1.columns(0, 2), from_vec(H, 2, 2, [1, 0, 0, 1, 2, 0, 0, 3], ctx2), 0.0)  # A broadcast
    assert result.to_array()[:, :, :2].tolist() == [[[1.0, 2.0], [4.0, 5.0]], [[2.0, 6.0], [8.0, 15.0]]]


def test_batched_incompatible_nbatch_is_an_error(H, ctx1, ctx2):  # :1435-1463 (#[should_panic])
    ctx3 = ctx1.clone_with_nbatch(3)
    z2, z3 = H.HipMat.zeros(2, 2, ctx2), H.HipMat.zeros(2, 2, ctx3)
    for a, b in ((z3, z2), (z2, z3)):
        with pytest.raises(H.DiffsolHipError) as e:
            H.HipMat.zeros(2, 2, ctx2).gemm(1.0, a, b, 0.0)
        assert e.value.code == -5
    with pytest.raises(H.DiffsolHipError) as e:
        z2.gemv(1.0, H.HipVec.zeros(2, ctx3), 0.0, H.HipVec.zeros(2, ctx2))
    assert e.value.code == -5


def test_batched_resize_cols(H, ctx2):  # :1465-1532
    a = from_vec(H, 2, 2, [1, 3, 2, 4, 5, 7, 6, 8], ctx2)
    a.resize_cols(3)
    assert (a.nrows, a.ncols) == (2, 3)
    assert a.to_array()[0].tolist() == [[1.0, 2.0, 0.0], [3.0, 4.0, 0.0]]
    y = H.HipVec.zeros(2, ctx2)
    a.gemv(1.0, vec(H, [1, 0, 0, 1, 0, 0], ctx2), 0.0, y)
    assert flat(y) == [1.0, 3.0, 5.0, 7.0]
    a.resize_cols(1)
    assert a.ncols == 1 and a.to_array()[0].tolist() == [[1.0], [3.0]]
    a.gemv(1.0, vec(H, [1, 1], ctx2), 0.0, y)
    assert flat(y) == [1.0, 3.0, 5.0, 7.0]


def test_batched_add_column_set_data_gather_mul_scalar(H, ctx1, ctx2):  # :1712-1863
    m = from_vec(H, 2, 2, [1, 2, 3, 4, 5, 6, 7, 8], ctx2)
    v = H.HipVec.zeros(2, ctx2)
    m.add_column_to_vector(1, v)
    assert flat(v) == [3.0, 4.0, 7.0, 8.0]
    z = H.HipMat.zeros(2, 2, ctx2)
    z.set_data_with_indices(H.HipIndex([0, 3], ctx1), H.HipIndex([0, 1], ctx1), vec(H, [5, 6, 50, 60], ctx2))
    assert vals(z) == [5.0, 0.0, 0.0, 6.0, 50.0, 0.0, 0.0, 60.0]
    m1 = from_vec(H, 3, 3, [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 20, 30, 40, 50, 60, 70, 80, 90], ctx2)
    m2 = H.HipMat.zeros(2, 2, ctx2)
    m2.gather(m1, H.HipIndex([0, 1, 3, 4], ctx1))
    assert vals(m2) == [1.0, 2.0, 4.0, 5.0, 10.0, 20.0, 40.0, 50.0]
    a = from_vec(H, 2, 2, [1, 3, 2, 4, 5, 7, 6, 8], ctx2)
    assert vals(a.mul_scalar(2.0)) == [2.0, 6.0, 4.0, 8.0, 10.0, 14.0, 12.0, 16.0]


def test_batched_partition_indices_and_column_axpy(H, ctx2):  # :1865-1902
    d = H.HipMat.from_diagonal(vec(H, [1, 0, 1, 2, 0, 2], ctx2))
    assert d.partition_indices_by_zero_diagonal() == ([1], [0, 2])
    a = from_vec(H, 2, 2, [1, 3, 2, 4, 5, 7, 6, 8], ctx2)
    a.column_axpy(2.0, 0, 1)
    assert a.to_array()[0].tolist() == [[1.0, 4.0], [3.0, 10.0]] and a.to_array()[1].tolist() == [[5.0, 16.0], [7.0, 22.0]]


def test_strided_column_views_of_a_larger_matrix(H, ctx1):  # make_strided_matrix + test_strided_matrix_view_* (:1962-2093): 3x4, nbatch 3
    nb, nr, nc = 3, 3, 4
    ctx3 = ctx1.clone_with_nbatch(nb)
    data = [r + c * 10.0 + b * 100.0 for b in range(nb) for c in range(nc) for r in range(nr)]
    m = from_vec(H, nr, nc, data, ctx3)
    owned = m.columns(1, 3).into_owned()
    ref = np.asarray(data).reshape(nb, nc, nr)[:, 1:3, :]
    assert np.array_equal(np.transpose(owned.to_array(), (0, 2, 1)), ref)
    doubled = owned.mul_scalar(2.0)
    assert np.array_equal(np.transpose(doubled.to_array(), (0, 2, 1)), 2.0 * ref)
    s = H.HipMat.zeros(nr, 2, ctx3)
    s.flat().copy_from(owned.flat().add(m.columns(0, 2).into_owned().flat()))
    assert np.array_equal(np.transpose(s.to_array(), (0, 2, 1)), ref + np.asarray(data).reshape(nb, nc, nr)[:, 0:2, :])


# ------------------------------------------------------------------ vector ops on matrix-column views (vector/mod.rs:1222-1540, test_strided_view_*)
def strided(H, ctx1, nb=2):
    ctx = ctx1.clone_with_nbatch(nb)
    data = np.asarray([r + c * 10.0 + b * 100.0 for b in range(nb) for c in range(4) for r in range(3)])
    return from_vec(H, 3, 4, data, ctx), data.reshape(nb, 4, 3), ctx  # ref[b][col][row]


def col(m, j):
    return m.column(j).clone_as_vec()  # [b][row]


def test_strided_view_set_index_copy_from_axpy_scale(H, ctx1):  # :1238-1304
    m, ref, ctx = strided(H, ctx1)
    m.column(1).set_index(1, 99.0)
    assert col(m, 1)[:, 1].tolist() == [99.0, 99.0] and np.array_equal(col(m, 0), ref[:, 0]) and np.array_equal(col(m, 2), ref[:, 2])
    m, ref, ctx = strided(H, ctx1)
    m.column(1).copy_from(vec(H, [50, 51, 52], ctx1))  # broadcast owned vector into the view
    assert col(m, 1).tolist() == [[50.0, 51.0, 52.0]] * 2
    m, ref, ctx = strided(H, ctx1)
    m.column(1).axpy(2.0, vec(H, [10, 10, 10], ctx1), 1.0)
    assert col(m, 1)[1].tolist() == [130.0, 131.0, 132.0] and col(m, 1)[0].tolist() == [30.0, 31.0, 32.0]
    m, ref, ctx = strided(H, ctx1)
    m.column(1).mul_assign(2.0)
    assert np.array_equal(col(m, 1), 2.0 * ref[:, 1]) and np.array_equal(col(m, 3), ref[:, 3])


def test_strided_view_add_sub_assign_and_broadcast(H, ctx1):  # :1306-1393
    m, ref, ctx = strided(H, ctx1)
    m.column(1).add_assign(m.column(2))  # view += view of the same matrix
    assert np.array_equal(col(m, 1), ref[:, 1] + ref[:, 2])
    m.column(3).sub_assign(m.column(0))
    assert np.array_equal(col(m, 3), ref[:, 3] - ref[:, 0])
    m, ref, ctx = strided(H, ctx1)
    m.column(2).add_assign(vec(H, [1, 2, 3], ctx1))  # broadcast operand
    assert np.array_equal(col(m, 2), ref[:, 2] + np.array([1.0, 2.0, 3.0]))
    s = m.column(0).add(vec(H, np.ones(6), ctx))  # view + owned -> owned
    assert np.array_equal(s.clone_as_vec(), ref[:, 0] + 1.0)


def test_strided_view_norms_component_ops_fill_and_index_ops(H, ctx1):  # :1395-1540
    m, ref, ctx = strided(H, ctx1)
    atol = vec(H, [1e-3, 1e-3, 1e-3], ctx1)
    y = vec(H, np.ones(6), ctx)
    expect = max(np.mean((ref[b, 1] / (1.0 * 0.1 + 1e-3)) ** 2) for b in range(2))
    assert np.isclose(m.column(1).squared_norm(y, atol, 0.1), expect, rtol=1e-14)
    assert np.array_equal(m.column(2).clone().clone_as_vec(), ref[:, 2])  # into_owned
    m.column(1).component_mul_assign(m.column(3))
    assert np.array_equal(col(m, 1), ref[:, 1] * ref[:, 3])
    m.column(3).component_div_assign(vec(H, [2, 4, 8], ctx1))
    assert np.array_equal(col(m, 3), ref[:, 3] / np.array([2.0, 4.0, 8.0]))
    assert np.array_equal(m.column(2).mul(3.0).clone_as_vec(), 3.0 * ref[:, 2])
    m.column(0).fill(7.0)
    assert (col(m, 0) == 7.0).all() and np.array_equal(col(m, 2), ref[:, 2])
    m, ref, ctx = strided(H, ctx1)
    m.column(1).assign_at_indices(H.HipIndex([0, 2], ctx1), -1.0)
    assert col(m, 1).tolist() == [[-1.0, 11.0, -1.0], [-1.0, 111.0, -1.0]]
    m.column(2).copy_from_indices(m.column(3), H.HipIndex([1], ctx1))
    assert col(m, 2).tolist() == [[20.0, 31.0, 22.0], [120.0, 131.0, 122.0]]
    g = H.HipVec.zeros(2, ctx)
    g.gather(m.column(3), H.HipIndex([2, 0], ctx1))
    assert g.clone_as_vec().tolist() == [[32.0, 30.0], [132.0, 130.0]]
    m, ref, ctx = strided(H, ctx1)
    result = H.HipVec.zeros(3, ctx)
    m.column(1).clone().scatter(H.HipIndex([0, 1, 2], ctx1), result)  # other[idx[i]] = self[i]
    assert result.clone_as_vec().tolist() == [[10.0, 11.0, 12.0], [110.0, 111.0, 112.0]]
    result2 = H.HipVec.zeros(3, ctx)
    m.column(3).scatter(H.HipIndex([2, 0], ctx1), result2)  # a view as the source, permuting indices
    assert result2.clone_as_vec().tolist() == [[31.0, 0.0, 30.0], [131.0, 0.0, 130.0]]
