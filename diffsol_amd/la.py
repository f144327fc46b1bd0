"""Thin Python mirror of the backend types (HipContext / HipVec / HipMat / HipLU) over the device C ABI — harness for the parity
tests of the Vector / Matrix / LinearSolver trait surface (reference: crates/diffsol-la/src/{vector,matrix,linear_solver}/mod.rs).
Host arrays use the reference's API layout: vectors [nbatch, n] (batch-major), matrices [nbatch, nrows, ncols]."""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import DiffsolHipError, check, vp


class HipContext:
    """Context (crates/diffsol-la/src/context/mod.rs:20-68): device + stream + nbatch."""

    def __init__(self, device=0, nbatch=1, stream=None, _share=None):
        self._L = _ffi.load_device_lib()
        self.nbatch = int(nbatch)
        if _share is not None:
            self._h = _share._h
            self._owner = _share._owner
        else:
            h = vp()
            check(self._L.dsh_ctx_create(device, stream, C.byref(h)))
            self._h = h
            self._owner = _CtxOwner(self._L, h)

    def clone_with_nbatch(self, nbatch):
        return HipContext(nbatch=nbatch, _share=self)

    def sync(self):
        check(self._L.dsh_ctx_sync(self._h))

    def set_block(self, threads):
        check(self._L.dsh_ctx_set_block(self._h, threads))


class _CtxOwner:
    def __init__(self, L, h):
        self.L, self.h, self.alive = L, h, True

    def __del__(self):
        # the cyclic collector may finalise a context before buffers that still point at it (e.g. objects kept alive by a traceback):
        # mark it dead so that their finalisers leave the (already released) device memory alone
        self.alive = False
        try:
            self.L.dsh_ctx_destroy(self.h)
        except Exception:
            pass


class _Buf:
    def __init__(self, ctx, nbytes, zero=True):
        self.ctx = ctx
        p = vp()
        check(ctx._L.dsh_malloc(ctx._h, nbytes, 1 if zero else 0, C.byref(p)))
        self.p = p

    def __del__(self):
        try:
            if self.ctx._owner.alive:
                self.ctx._L.dsh_free(self.ctx._h, self.p)
        except Exception:
            pass


def _ptr(buf, offset_elems=0):
    return vp(buf.p.value + 8 * int(offset_elems))


class HipIndex:
    def __init__(self, idx, ctx):
        self.host = np.ascontiguousarray(idx, dtype=np.int32)
        self.ctx = ctx
        self._buf = _Buf(ctx, max(4 * self.host.size, 4))
        if self.host.size:
            check(ctx._L.dsh_h2d(ctx._h, self._buf.p, self.host.ctypes.data_as(vp), 4 * self.host.size))

    def __len__(self):
        return int(self.host.size)


class HipVec:
    """Vector (crates/diffsol-la/src/vector/mod.rs:163-377) for the HIP backend; a view is (buffer, offset, n)."""

    def __init__(self, ctx, n, _buf=None, _off=0, zero=True):
        self.ctx, self.n = ctx, int(n)
        self._buf = _buf if _buf is not None else _Buf(ctx, max(8 * self.n * ctx.nbatch, 8), zero)
        self._off = _off

    # -- construction / transfer
    @staticmethod
    def zeros(n, ctx):
        return HipVec(ctx, n)

    @staticmethod
    def from_element(n, value, ctx):
        v = HipVec(ctx, n, zero=False)
        v.fill(value)
        return v

    @staticmethod
    def from_vec(data, ctx):
        """data: [nbatch, n] or flat batch-major (vector/cuda.rs:741-760)."""
        a = np.ascontiguousarray(data, dtype=np.float64).reshape(-1)
        if a.size % ctx.nbatch:
            raise DiffsolHipError(-1, "from_vec: length must be a multiple of nbatch")
        v = HipVec(ctx, a.size // ctx.nbatch, zero=False)
        check(ctx._L.dsh_vec_upload(ctx._h, v.n, ctx.nbatch, a.ctypes.data_as(_ffi.c_dp), v.ptr))
        return v

    def clone_as_vec(self):
        out = np.empty((self.nb, self.n))
        check(self.ctx._L.dsh_vec_download(self.ctx._h, self.n, self.nb, self.ptr, out.ctypes.data_as(_ffi.c_dp)))
        return out

    def clone(self):
        v = HipVec(self.ctx, self.n, zero=False)
        v.copy_from(self)
        return v

    @property
    def ptr(self):
        return _ptr(self._buf, self._off)

    @property
    def nb(self):
        return self.ctx.nbatch

    def __len__(self):
        return self.n

    def _compat(self, o):
        if o.n != self.n:
            raise DiffsolHipError(-1, f"Vector length mismatch: {self.n} vs {o.n}")
        if o.nb not in (1, self.nb):
            raise DiffsolHipError(-1, f"Incompatible nbatch: {self.nb} vs {o.nb}")

    # -- ops (each one dsh_* call)
    def fill(self, v):
        check(self.ctx._L.dsh_vec_fill(self.ctx._h, self.n, self.nb, self.ptr, v))

    def copy_from(self, o):
        self._compat(o)
        check(self.ctx._L.dsh_vec_copy(self.ctx._h, self.n, self.nb, o.ptr, o.nb, self.ptr))

    def axpy(self, alpha, x, beta):
        self._compat(x)
        check(self.ctx._L.dsh_vec_axpy(self.ctx._h, self.n, self.nb, alpha, x.ptr, x.nb, beta, self.ptr))

    def batched_axpy(self, alpha, x, beta):
        self._compat(x)
        a = np.ascontiguousarray(alpha, dtype=np.float64)
        if a.size != self.nb:
            raise DiffsolHipError(-1, "batched_axpy: alpha must have nbatch entries")
        check(self.ctx._L.dsh_vec_batched_axpy(self.ctx._h, self.n, self.nb, a.ctypes.data_as(_ffi.c_dp), x.ptr, x.nb, beta, self.ptr))

    def add_assign(self, o):
        self._compat(o)
        check(self.ctx._L.dsh_vec_add_assign(self.ctx._h, self.n, self.nb, self.ptr, o.ptr, o.nb))

    def sub_assign(self, o):
        self._compat(o)
        check(self.ctx._L.dsh_vec_sub_assign(self.ctx._h, self.n, self.nb, self.ptr, o.ptr, o.nb))

    def component_mul_assign(self, o):
        self._compat(o)
        check(self.ctx._L.dsh_vec_mul_assign(self.ctx._h, self.n, self.nb, self.ptr, o.ptr, o.nb))

    def component_div_assign(self, o):
        self._compat(o)
        check(self.ctx._L.dsh_vec_div_assign(self.ctx._h, self.n, self.nb, self.ptr, o.ptr, o.nb))

    def mul_assign(self, s):
        check(self.ctx._L.dsh_vec_mul_assign_scalar(self.ctx._h, self.n, self.nb, self.ptr, s))

    def add(self, o):
        self._compat(o)
        r = HipVec(self.ctx, self.n, zero=False)
        check(self.ctx._L.dsh_vec_add(self.ctx._h, self.n, self.nb, self.ptr, self.nb, o.ptr, o.nb, r.ptr))
        return r

    def sub(self, o):
        self._compat(o)
        r = HipVec(self.ctx, self.n, zero=False)
        check(self.ctx._L.dsh_vec_sub(self.ctx._h, self.n, self.nb, self.ptr, self.nb, o.ptr, o.nb, r.ptr))
        return r

    def mul(self, s):
        r = HipVec(self.ctx, self.n, zero=False)
        check(self.ctx._L.dsh_vec_mul_scalar(self.ctx._h, self.n, self.nb, self.ptr, s, r.ptr))
        return r

    def norm(self, k):
        out = C.c_double()
        check(self.ctx._L.dsh_vec_norm(self.ctx._h, self.n, self.nb, self.ptr, k, C.byref(out)))
        return out.value

    def squared_norm(self, y, atol, rtol, per_batch=False):
        if y.n != self.n or atol.n != self.n:
            raise DiffsolHipError(-1, "Vector lengths do not match")
        out = C.c_double()
        pb = HipVec(self.ctx.clone_with_nbatch(self.nb), 1) if per_batch else None
        check(self.ctx._L.dsh_vec_squared_norm(self.ctx._h, self.n, self.nb, self.ptr, y.ptr, y.nb, atol.ptr, atol.nb, rtol, C.byref(out),
                                               pb.ptr if pb else None))
        if per_batch:
            return out.value, pb.clone_as_vec().reshape(-1)
        return out.value

    def get_index(self, i):
        if self.nb != 1:
            raise DiffsolHipError(-1, "get_index is only valid for nbatch == 1")
        if not 0 <= i < self.n:
            raise DiffsolHipError(-1, "index out of bounds")
        out = C.c_double()
        check(self.ctx._L.dsh_vec_get_index(self.ctx._h, 1, self.ptr, i, 0, C.byref(out)))
        return out.value

    def get_batch(self, b):
        """Vector::get_batch (vector/mod.rs:227): member b as an owned vector with nbatch = 1 (dsh_vec_extract_batch)."""
        out = HipVec(self.ctx.clone_with_nbatch(1), self.n)
        check(self.ctx._L.dsh_vec_extract_batch(self.ctx._h, self.n, self.nb, self.ptr, b, out.ptr))
        return out

    def set_batch(self, b, v):
        """write-back of a get_batch_mut view: member b <- v (nbatch 1)"""
        if v.n != self.n or v.nb != 1:
            raise DiffsolHipError(-1, "set_batch needs a vector of the same length with nbatch == 1")
        check(self.ctx._L.dsh_vec_insert_batch(self.ctx._h, self.n, self.nb, self.ptr, b, v.ptr))

    def set_index(self, i, v):
        if not 0 <= i < self.n:
            raise DiffsolHipError(-1, "index out of bounds")
        check(self.ctx._L.dsh_vec_set_index_all(self.ctx._h, self.nb, self.ptr, i, v))

    def gather(self, other, idx):
        check(self.ctx._L.dsh_vec_gather(self.ctx._h, other.n, self.nb, other.ptr, idx._buf.p, len(idx), self.ptr))

    def scatter(self, idx, other):
        check(self.ctx._L.dsh_vec_scatter(self.ctx._h, other.n, self.nb, self.ptr, idx._buf.p, len(idx), other.ptr))

    def copy_from_indices(self, other, idx):
        check(self.ctx._L.dsh_vec_copy_from_indices(self.ctx._h, self.n, self.nb, other.ptr, idx._buf.p, len(idx), self.ptr))

    def assign_at_indices(self, idx, value):
        check(self.ctx._L.dsh_vec_assign_at_indices(self.ctx._h, self.n, self.nb, idx._buf.p, len(idx), value, self.ptr))

    def root_finding(self, g1):
        found, idx, frac = C.c_int(), C.c_int(), C.c_double()
        check(self.ctx._L.dsh_vec_root_finding(self.ctx._h, self.n, self.nb, self.ptr, g1.ptr, C.byref(found), C.byref(frac), C.byref(idx)))
        return bool(found.value), frac.value, idx.value


class HipMat:
    """DenseMatrix (crates/diffsol-la/src/matrix/mod.rs:169-424): column-major per system, batch-fastest on the device."""

    def __init__(self, ctx, nrows, ncols, zero=True):
        self.ctx, self.nrows, self.ncols = ctx, int(nrows), int(ncols)
        self._buf = _Buf(ctx, max(8 * self.nrows * self.ncols * ctx.nbatch, 8), zero)

    @staticmethod
    def zeros(nrows, ncols, ctx):
        return HipMat(ctx, nrows, ncols)

    @staticmethod
    def from_array(a, ctx):
        """a: [nbatch, nrows, ncols] (or [nrows, ncols] for nbatch 1)."""
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 2:
            a = a[None]
        nb, nr, nc = a.shape
        if nb != ctx.nbatch:
            raise DiffsolHipError(-1, "from_array: leading dimension must be nbatch")
        m = HipMat(ctx, nr, nc, zero=False)
        cm = np.ascontiguousarray(np.transpose(a, (0, 2, 1)))  # [b][col][row]
        check(ctx._L.dsh_vec_upload(ctx._h, nr * nc, nb, cm.ctypes.data_as(_ffi.c_dp), m.ptr))
        return m

    @staticmethod
    def from_diagonal(v):
        m = HipMat(v.ctx, v.n, v.n, zero=False)
        check(v.ctx._L.dsh_mat_from_diagonal(v.ctx._h, v.n, v.nb, v.ptr, v.nb, m.ptr))
        return m

    def to_array(self):
        out = np.empty((self.nb, self.ncols, self.nrows))
        check(self.ctx._L.dsh_vec_download(self.ctx._h, self.nrows * self.ncols, self.nb, self.ptr, out.ctypes.data_as(_ffi.c_dp)))
        return np.transpose(out, (0, 2, 1)).copy()

    @property
    def ptr(self):
        return _ptr(self._buf)

    @property
    def nb(self):
        return self.ctx.nbatch

    def column(self, j):
        if not 0 <= j < self.ncols:
            raise DiffsolHipError(-1, "Column index out of bounds")
        return HipVec(self.ctx, self.nrows, _buf=self._buf, _off=j * self.nrows * self.nb)

    def diagonal(self):
        v = HipVec(self.ctx, self.nrows)
        check(self.ctx._L.dsh_mat_get_diagonal(self.ctx._h, self.nrows, self.nb, self.ptr, v.ptr))
        return v

    def set_column(self, j, v):
        check(self.ctx._L.dsh_mat_set_column(self.ctx._h, self.nrows, self.ncols, self.nb, self.ptr, j, v.ptr, v.nb))

    def scale_add_and_assign(self, x, beta, y):
        check(self.ctx._L.dsh_mat_scale_add_assign(self.ctx._h, self.nrows * self.ncols, self.nb, self.ptr, x.ptr, x.nb, beta, y.ptr, y.nb))

    def column_axpy(self, alpha, j, i):
        if not (0 <= i < self.ncols and 0 <= j < self.ncols):
            raise DiffsolHipError(-1, "Column index out of bounds")
        check(self.ctx._L.dsh_mat_column_axpy(self.ctx._h, self.nrows, self.nb, self.ptr, alpha, j, i))

    def gemv(self, alpha, x, beta, y):
        if x.n != self.ncols or y.n != self.nrows:
            raise DiffsolHipError(-1, "gemv: shape mismatch")
        check(self.ctx._L.dsh_mat_gemv(self.ctx._h, self.nrows, self.ncols, y.nb, alpha, self.ptr, self.nb, x.ptr, x.nb, beta, y.ptr))

    def gemm(self, alpha, a, b, beta):
        if a.nrows != self.nrows or b.ncols != self.ncols or a.ncols != b.nrows:
            raise DiffsolHipError(-1, "gemm: shape mismatch")
        check(self.ctx._L.dsh_mat_gemm(self.ctx._h, self.nrows, self.ncols, a.ncols, self.nb, alpha, a.ptr, a.nb, b.ptr, b.nb, beta, self.ptr))

    # -- the rest of the DenseMatrix surface, composed from the same C-ABI entry points the Rust shim would use
    def flat(self):
        """All entries as one batched vector (column after column): Matrix ops that are element-wise act on this view."""
        return HipVec(self.ctx, self.nrows * self.ncols, _buf=self._buf, _off=0)

    def copy_from(self, other):
        self.flat().copy_from(other.flat())

    def columns(self, start, end):
        """DenseMatrix::columns(start, end): a non-owning view of columns [start, end) — contiguous in the batch-fastest layout."""
        if not 0 <= start <= end <= self.ncols:
            raise DiffsolHipError(-1, "Column range out of bounds")
        return HipMatView(self, start, end)

    def add_column_to_vector(self, j, v):
        v.add_assign(self.column(j))

    def mul_scalar(self, s):
        out = HipMat(self.ctx, self.nrows, self.ncols, zero=False)
        out.flat().copy_from(self.flat().mul(s))
        return out

    def gather(self, other, idx):
        self.flat().gather(other.flat(), idx)

    def mat_mul(self, b):
        ctx = self.ctx if self.nb >= b.nb else b.ctx
        c = HipMat(ctx, self.nrows, b.ncols)
        c.gemm(1.0, self, b, 0.0)
        return c

    def resize_cols(self, ncols):
        """DenseMatrix::resize_cols (matrix/mod.rs:399-402): keep the leading columns, zero-fill new ones."""
        new = HipMat(self.ctx, self.nrows, ncols)
        keep = min(ncols, self.ncols)
        if keep:
            HipVec(self.ctx, self.nrows * keep, _buf=new._buf).copy_from(HipVec(self.ctx, self.nrows * keep, _buf=self._buf))
        self._buf, self.ncols = new._buf, int(ncols)

    def partition_indices_by_zero_diagonal(self):
        """Matrix::partition_indices_by_zero_diagonal (matrix/mod.rs:319-330): decided on batch member 0 like the reference."""
        d = self.diagonal().clone_as_vec().reshape(self.nb, self.nrows)[0]
        return [i for i in range(self.nrows) if d[i] == 0.0], [i for i in range(self.nrows) if d[i] != 0.0]

    def set_data_with_indices(self, dst_idx, src_idx, data):
        check(self.ctx._L.dsh_mat_set_data_with_indices(self.ctx._h, self.nrows * self.ncols, data.n, self.nb, self.ptr, dst_idx._buf.p, src_idx._buf.p,
                                                        len(dst_idx), data.ptr))


class HipMatView:
    """MatrixView over a column range (matrix/mod.rs:98-167): (parent buffer, first column, ncols)."""

    def __init__(self, parent, start, end):
        self.parent, self.start, self.ncols, self.nrows, self.ctx = parent, int(start), int(end - start), parent.nrows, parent.ctx

    @property
    def ptr(self):
        return self.parent.column(self.start).ptr if self.ncols else self.parent.ptr

    @property
    def nb(self):
        return self.parent.nb

    def into_owned(self):
        m = HipMat(self.ctx, self.nrows, self.ncols, zero=False)
        HipVec(self.ctx, self.nrows * self.ncols, _buf=m._buf).copy_from(HipVec(self.ctx, self.nrows * self.ncols, _buf=self.parent._buf,
                                                                                _off=self.start * self.nrows * self.nb))
        return m

    def gemv_o(self, alpha, x, beta, y):
        if x.n != self.ncols or y.n != self.nrows:
            raise DiffsolHipError(-1, "gemv: shape mismatch")
        check(self.ctx._L.dsh_mat_gemv(y.ctx._h, self.nrows, self.ncols, y.nb, alpha, self.ptr, self.nb, x.ptr, x.nb, beta, y.ptr))

    gemv_v = gemv_o

    def gemm_vo(self, alpha, a_view, b, beta):
        """self (a mutable column-range view) = alpha * a_view * b + beta * self"""
        if a_view.nrows != self.nrows or b.ncols != self.ncols or a_view.ncols != b.nrows:
            raise DiffsolHipError(-1, "gemm: shape mismatch")
        check(self.ctx._L.dsh_mat_gemm(self.ctx._h, self.nrows, self.ncols, a_view.ncols, self.nb, alpha, a_view.ptr, a_view.nb, b.ptr, b.nb, beta, self.ptr))

    gemm_oo = gemm_vo


class HipLU:
    """LinearSolver (crates/diffsol-la/src/linear_solver/mod.rs:19-42) — batched dense LU, one launch per factor / solve."""

    def __init__(self, ctx, n):
        self.ctx, self.n = ctx, int(n)
        h = vp()
        check(ctx._L.dsh_lu_create(ctx._h, n, ctx.nbatch, C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if self.ctx._owner.alive:
                self.ctx._L.dsh_lu_destroy(self._h)
        except Exception:
            pass

    def factor(self, mat):
        if mat.nrows != self.n or mat.ncols != self.n:
            raise DiffsolHipError(-1, "LinearSolverMatrixNotSquare")
        check(self.ctx._L.dsh_lu_factor(self._h, mat.ptr))

    def solve_in_place(self, v):
        if v.n != self.n:
            raise DiffsolHipError(-1, "LinearSolverMatrixVectorNotCompatible")
        check(self.ctx._L.dsh_lu_solve(self._h, v.ptr))

    def set_structure(self, dense):
        """DSH_LU_STRUCTURE_DENSE (True) / _AUTO (False): banded operands in dense containers are detected and solved by the banded kernels."""
        check(self.ctx._L.dsh_lu_set_structure(self._h, 1 if dense else 0))

    def band_width(self):
        return int(self.ctx._L.dsh_lu_band_width(self._h))

    def n_singular(self):
        out = C.c_int64()
        check(self.ctx._L.dsh_lu_info(self._h, C.byref(out)))
        return out.value

    def factors(self):
        """[nbatch, n, n] packed LU and [nbatch, n] pivot rows."""
        lu = np.empty((self.ctx.nbatch, self.n, self.n))  # [b][col][row]
        piv = np.empty((self.ctx.nbatch, self.n), dtype=np.int32)
        check(self.ctx._L.dsh_lu_download(self._h, lu.ctypes.data_as(_ffi.c_dp), piv.ctypes.data_as(_ffi.c_i32p)))
        return np.transpose(lu, (0, 2, 1)).copy(), piv
