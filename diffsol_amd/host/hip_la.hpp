// Host-side mirror of diffsol-la's trait surface over the C ABI of libdiffsol_hip.so.
//
// The reference's backend seam is a set of Rust types implementing Context / Vector / VectorView / Matrix / DenseMatrix /
// LinearSolver (diffsol-la/src/lib.rs:72-112).  There is no Rust toolchain in this environment, so the same surface is written
// in C++ with the reference's names and argument meaning; every method forwards to exactly one `dsh_*` entry point, the way the
// Rust `HipVec / HipMat / HipLU` shim in INTEGRATION.md does.  Shape / nbatch mismatches that panic in the reference throw
// `LaError` here.
//   Context      diffsol-la/src/context/mod.rs:20-68      -> HipContext
//   Vector       diffsol-la/src/vector/mod.rs:163-377     -> HipVec (+ HipVecView: VectorView :140-156)
//   DenseMatrix  diffsol-la/src/matrix/mod.rs:169-424     -> HipMat
//   LinearSolver diffsol-la/src/linear_solver/mod.rs:19-42 -> HipLU
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/diffsol_hip.h"

namespace diffsol_hip {

struct LaError : std::runtime_error {
  int code;
  LaError(int code_, const std::string& what) : std::runtime_error(what), code(code_) {}
};

inline void check(int rc, const char* where) {
  if (rc != DSH_OK) throw LaError(rc, std::string(where) + ": " + dsh_last_error());
}

// Context: device + stream + nbatch (CudaContext, context/cuda.rs:41-144).  Cheap to copy (shared handle).
class HipContext {
 public:
  HipContext() = default;
  explicit HipContext(int device, void* stream = nullptr, int64_t nbatch = 1) : nbatch_(nbatch) {
    dsh_ctx* c = nullptr;
    check(dsh_ctx_create(device, stream, &c), "HipContext");
    ctx_ = std::shared_ptr<dsh_ctx>(c, [](dsh_ctx* p) { dsh_ctx_destroy(p); });
  }
  int64_t nbatch() const { return nbatch_; }
  // Context::clone_with_nbatch (context/mod.rs:56-67)
  HipContext clone_with_nbatch(int64_t nb) const {
    if (nb < 1) throw LaError(DSH_E_INVALID, "nbatch must be >= 1");
    HipContext c = *this;
    c.nbatch_ = nb;
    return c;
  }
  dsh_ctx* raw() const { return ctx_.get(); }
  bool valid() const { return (bool)ctx_; }
  void sync() const { check(dsh_ctx_sync(raw()), "sync"); }
  // Context::check_compatible (context/mod.rs:28-54): operands must have equal nbatch or nbatch 1
  static void check_compatible(int64_t self_nb, int64_t other_nb) {
    if (other_nb != 1 && other_nb != self_nb) throw LaError(DSH_E_INVALID, "Incompatible nbatch: " + std::to_string(self_nb) + " vs " + std::to_string(other_nb));
  }

 private:
  std::shared_ptr<dsh_ctx> ctx_;
  int64_t nbatch_ = 1;
};

// Scale<T> wrapper (scalar/mod.rs:153-216)
struct Scale { double v; };
inline Scale scale(double v) { return Scale{v}; }

class HipVec;

// Non-owning batched vector view: (ptr, nstates, nbatch) — a column / column range of a HipMat or a whole HipVec
// (CudaVecRef / CudaVecMut, vector/cuda.rs:148-207; with the batch-fastest layout a column view is a plain pointer offset).
struct HipVecView {
  const double* p = nullptr;
  int64_t n = 0;
  int64_t nb = 1;
  HipContext ctx;
  int64_t len() const { return n; }
  double squared_norm(const HipVec& y, const HipVec& atol, double rtol) const;
  HipVec into_owned() const;
};
struct HipVecViewMut {
  double* p = nullptr;
  int64_t n = 0;
  int64_t nb = 1;
  HipContext ctx;
  int64_t len() const { return n; }
  void copy_from(const HipVec& x);
  void copy_from_view(const HipVecView& x);
  void axpy(double alpha, const HipVec& x, double beta);
  void mul_assign(Scale s) { check(dsh_vec_mul_assign_scalar(ctx.raw(), n, nb, p, s.v), "mul_assign"); }
  HipVecView as_view() const { return HipVecView{p, n, nb, ctx}; }
};

// Device index vector (CudaIndex, vector/cuda.rs:135-138): int32 indices shared by all batch members
class HipIndex {
 public:
  HipIndex() = default;
  HipIndex(const std::vector<int>& idx, const HipContext& ctx) : ctx_(ctx), host_(idx) {
    void* d = nullptr;
    check(dsh_malloc(ctx.raw(), (int64_t)(sizeof(int32_t) * (idx.size() ? idx.size() : 1)), 0, &d), "HipIndex");
    dsh_ctx* c = ctx.raw();
    std::shared_ptr<dsh_ctx> keep = nullptr;
    dev_ = std::shared_ptr<int32_t>((int32_t*)d, [ctx](int32_t* p) { dsh_free(ctx.raw(), p); });
    (void)c;
    std::vector<int32_t> h(idx.begin(), idx.end());
    if (!h.empty()) check(dsh_h2d(ctx.raw(), d, h.data(), (int64_t)(sizeof(int32_t) * h.size())), "HipIndex");
  }
  int64_t len() const { return (int64_t)host_.size(); }
  const int32_t* dev() const { return dev_.get(); }
  const std::vector<int>& host() const { return host_; }

 private:
  HipContext ctx_;
  std::vector<int> host_;
  std::shared_ptr<int32_t> dev_;
};

// Owning batched vector (CudaVec, vector/cuda.rs:127-146, Vector impl :741-1309)
class HipVec {
 public:
  HipVec() = default;
  static HipVec zeros(int64_t nstates, const HipContext& ctx) { return HipVec(nstates, ctx, true); }
  static HipVec from_element(int64_t nstates, double value, const HipContext& ctx) {
    HipVec v(nstates, ctx, false);
    v.fill(value);
    return v;
  }
  // data is batch-major: [b0 states..., b1 states...]  (vector/cuda.rs:741-760)
  static HipVec from_vec(const std::vector<double>& data, const HipContext& ctx) {
    int64_t nb = ctx.nbatch();
    if ((int64_t)data.size() % nb != 0) throw LaError(DSH_E_INVALID, "from_vec: length must be a multiple of nbatch");
    HipVec v((int64_t)data.size() / nb, ctx, false);
    check(dsh_vec_upload(ctx.raw(), v.n_, nb, data.data(), v.ptr()), "from_vec");
    return v;
  }
  std::vector<double> clone_as_vec() const {
    std::vector<double> out((size_t)(n_ * nb()));
    check(dsh_vec_download(ctx_.raw(), n_, nb(), ptr(), out.data()), "clone_as_vec");
    return out;
  }
  HipVec clone() const {
    HipVec v(n_, ctx_, false);
    v.copy_from(*this);
    return v;
  }
  int64_t len() const { return n_; }
  int64_t nb() const { return ctx_.nbatch(); }
  const HipContext& context() const { return ctx_; }
  double* ptr() { return data_.get(); }
  const double* ptr() const { return data_.get(); }
  HipVecView as_view() const { return HipVecView{ptr(), n_, nb(), ctx_}; }
  HipVecViewMut as_view_mut() { return HipVecViewMut{ptr(), n_, nb(), ctx_}; }

  void fill(double v) { check(dsh_vec_fill(ctx_.raw(), n_, nb(), ptr(), v), "fill"); }
  void copy_from(const HipVec& x) { same_len(x.n_); HipContext::check_compatible(nb(), x.nb()); check(dsh_vec_copy(ctx_.raw(), n_, nb(), x.ptr(), x.nb(), ptr()), "copy_from"); }
  void copy_from_view(const HipVecView& x) { same_len(x.n); HipContext::check_compatible(nb(), x.nb); check(dsh_vec_copy(ctx_.raw(), n_, nb(), x.p, x.nb, ptr()), "copy_from_view"); }
  // y = alpha*x + beta*y (vector/cuda.rs:937-966)
  // self = alpha*x + beta*y0 (copy_from(y0) + axpy in one pass); copy_x_to (optional, same shape) also receives x
  void assign_axpby(double alpha, const double* x, double beta, const double* y0, double* copy_x_to = nullptr) {
    check(dsh_vec_axpby_to(ctx_.raw(), n_, nb(), alpha, x, beta, y0, ptr(), copy_x_to), "assign_axpby");
  }
  void axpy(double alpha, const HipVec& x, double beta) { same_len(x.n_); HipContext::check_compatible(nb(), x.nb()); check(dsh_vec_axpy(ctx_.raw(), n_, nb(), alpha, x.ptr(), x.nb(), beta, ptr()), "axpy"); }
  void axpy_v(double alpha, const HipVecView& x, double beta) { same_len(x.n); HipContext::check_compatible(nb(), x.nb); check(dsh_vec_axpy(ctx_.raw(), n_, nb(), alpha, x.p, x.nb, beta, ptr()), "axpy_v"); }
  void batched_axpy(const std::vector<double>& alpha, const HipVec& x, double beta) {
    same_len(x.n_);
    if ((int64_t)alpha.size() != nb()) throw LaError(DSH_E_INVALID, "batched_axpy: alpha must have nbatch entries");
    check(dsh_vec_batched_axpy(ctx_.raw(), n_, nb(), alpha.data(), x.ptr(), x.nb(), beta, ptr()), "batched_axpy");
  }
  void add_assign(const HipVec& x) { same_len(x.n_); HipContext::check_compatible(nb(), x.nb()); check(dsh_vec_add_assign(ctx_.raw(), n_, nb(), ptr(), x.ptr(), x.nb()), "add_assign"); }
  void add_assign(const HipVecView& x) { same_len(x.n); HipContext::check_compatible(nb(), x.nb); check(dsh_vec_add_assign(ctx_.raw(), n_, nb(), ptr(), x.p, x.nb), "add_assign"); }
  void sub_assign(const HipVec& x) { same_len(x.n_); HipContext::check_compatible(nb(), x.nb()); check(dsh_vec_sub_assign(ctx_.raw(), n_, nb(), ptr(), x.ptr(), x.nb()), "sub_assign"); }
  void sub_assign(const HipVecView& x) { same_len(x.n); HipContext::check_compatible(nb(), x.nb); check(dsh_vec_sub_assign(ctx_.raw(), n_, nb(), ptr(), x.p, x.nb), "sub_assign"); }
  void mul_assign(Scale s) { check(dsh_vec_mul_assign_scalar(ctx_.raw(), n_, nb(), ptr(), s.v), "mul_assign"); }
  void component_mul_assign(const HipVec& x) { same_len(x.n_); HipContext::check_compatible(nb(), x.nb()); check(dsh_vec_mul_assign(ctx_.raw(), n_, nb(), ptr(), x.ptr(), x.nb()), "component_mul_assign"); }
  void component_div_assign(const HipVec& x) { same_len(x.n_); HipContext::check_compatible(nb(), x.nb()); check(dsh_vec_div_assign(ctx_.raw(), n_, nb(), ptr(), x.ptr(), x.nb()), "component_div_assign"); }
  // ret = self + x / self - x  (ops :439-484)
  HipVec add(const HipVec& x) const { HipVec r(n_, ctx_, false); check(dsh_vec_add(ctx_.raw(), n_, nb(), ptr(), nb(), x.ptr(), x.nb(), r.ptr()), "add"); return r; }
  HipVec sub(const HipVec& x) const { HipVec r(n_, ctx_, false); check(dsh_vec_sub(ctx_.raw(), n_, nb(), ptr(), nb(), x.ptr(), x.nb(), r.ptr()), "sub"); return r; }
  HipVec mul(Scale s) const { HipVec r(n_, ctx_, false); check(dsh_vec_mul_scalar(ctx_.raw(), n_, nb(), ptr(), s.v, r.ptr()), "mul"); return r; }

  // max over batches of mean_i (x_i/(|y_i| rtol + atol_i))^2   (vector/cuda.rs:1362-1433)
  double squared_norm(const HipVec& y, const HipVec& atol, double rtol) const { return as_view().squared_norm(y, atol, rtol); }
  double norm(int k) const { double out = 0.0; check(dsh_vec_norm(ctx_.raw(), n_, nb(), ptr(), k, &out), "norm"); return out; }
  // get_index panics when nbatch > 1 (vector/mod.rs test_batched_get_index_panics)
  double get_index(int64_t i) const {
    if (nb() != 1) throw LaError(DSH_E_INVALID, "get_index is only valid for nbatch == 1");
    if (i < 0 || i >= n_) throw LaError(DSH_E_INVALID, "index out of bounds");
    double out = 0.0;
    check(dsh_vec_get_index(ctx_.raw(), 1, ptr(), i, 0, &out), "get_index");
    return out;
  }
  // set_index sets element i of every batch member (vector/cuda.rs:762-774)
  void set_index(int64_t i, double v) {
    if (i < 0 || i >= n_) throw LaError(DSH_E_INVALID, "index out of bounds");
    check(dsh_vec_set_index_all(ctx_.raw(), nb(), ptr(), i, v), "set_index");
  }
  void gather(const HipVec& other, const HipIndex& idx) { if (idx.len() != n_) throw LaError(DSH_E_INVALID, "gather: index length != self length"); check(dsh_vec_gather(ctx_.raw(), other.n_, nb(), other.ptr(), idx.dev(), idx.len(), ptr()), "gather"); }
  void scatter(const HipIndex& idx, HipVec& other) const { if (idx.len() != n_) throw LaError(DSH_E_INVALID, "scatter: index length != self length"); check(dsh_vec_scatter(ctx_.raw(), other.n_, nb(), ptr(), idx.dev(), idx.len(), other.ptr()), "scatter"); }
  void copy_from_indices(const HipVec& other, const HipIndex& idx) { same_len(other.n_); check(dsh_vec_copy_from_indices(ctx_.raw(), n_, nb(), other.ptr(), idx.dev(), idx.len(), ptr()), "copy_from_indices"); }
  void assign_at_indices(const HipIndex& idx, double v) { check(dsh_vec_assign_at_indices(ctx_.raw(), n_, nb(), idx.dev(), idx.len(), v, ptr()), "assign_at_indices"); }
  // (found_root, max_frac, max_frac_index) of batch member 0; throws if batches disagree (vector/cuda.rs:1153-1177 panics)
  void root_finding(const HipVec& g1, bool& found, double& frac, int& idx) const {
    same_len(g1.n_);
    int f = 0;
    check(dsh_vec_root_finding(ctx_.raw(), n_, nb(), ptr(), g1.ptr(), &f, &frac, &idx), "root_finding");
    found = f != 0;
  }

 private:
  HipVec(int64_t n, const HipContext& ctx, bool zero) : n_(n), ctx_(ctx) {
    void* d = nullptr;
    check(dsh_malloc(ctx.raw(), (int64_t)sizeof(double) * n * ctx.nbatch(), zero ? 1 : 0, &d), "HipVec alloc");
    data_ = std::shared_ptr<double>((double*)d, [ctx](double* p) { dsh_free(ctx.raw(), p); });
  }
  void same_len(int64_t other) const { if (other != n_) throw LaError(DSH_E_INVALID, "Vector length mismatch: " + std::to_string(n_) + " vs " + std::to_string(other)); }
  int64_t n_ = 0;
  HipContext ctx_;
  std::shared_ptr<double> data_;
};

inline double HipVecView::squared_norm(const HipVec& y, const HipVec& atol, double rtol) const {
  if (y.len() != n || atol.len() != n) throw LaError(DSH_E_INVALID, "Vector lengths do not match");
  HipContext::check_compatible(nb, y.nb());
  HipContext::check_compatible(nb, atol.nb());
  double out = 0.0;
  check(dsh_vec_squared_norm(ctx.raw(), n, nb, p, y.ptr(), y.nb(), atol.ptr(), atol.nb(), rtol, &out, nullptr), "squared_norm");
  return out;
}
inline HipVec HipVecView::into_owned() const {
  HipVec v = HipVec::zeros(n, ctx.clone_with_nbatch(nb));
  v.copy_from_view(*this);
  return v;
}
inline void HipVecViewMut::copy_from(const HipVec& x) {
  if (x.len() != n) throw LaError(DSH_E_INVALID, "Vector length mismatch");
  check(dsh_vec_copy(ctx.raw(), n, nb, x.ptr(), x.nb(), p), "copy_from");
}
inline void HipVecViewMut::copy_from_view(const HipVecView& x) {
  if (x.n != n) throw LaError(DSH_E_INVALID, "Vector length mismatch");
  check(dsh_vec_copy(ctx.raw(), n, nb, x.p, x.nb, p), "copy_from_view");
}
inline void HipVecViewMut::axpy(double alpha, const HipVec& x, double beta) {
  if (x.len() != n) throw LaError(DSH_E_INVALID, "Vector length mismatch");
  check(dsh_vec_axpy(ctx.raw(), n, nb, alpha, x.ptr(), x.nb(), beta, p), "axpy");
}

// Column-range view of a dense matrix (CudaMatRef, matrix/cuda.rs:43-62)
struct HipMatView {
  const double* p = nullptr;
  int64_t nrows = 0, ncols = 0, nb = 1;
  HipContext ctx;
  // y = alpha*self*x + beta*y  (gemv_o / gemv_v, matrix/cuda.rs:620-677)
  void gemv_o(double alpha, const HipVec& x, double beta, HipVec& y) const {
    if (x.len() != ncols || y.len() != nrows) throw LaError(DSH_E_INVALID, "gemv: shape mismatch");
    check(dsh_mat_gemv(ctx.raw(), nrows, ncols, nb, alpha, p, nb, x.ptr(), x.nb(), beta, y.ptr()), "gemv_o");
  }
  // y = alpha*self*x + beta*y0: y.copy_from(y0) + gemv_o in one pass
  void gemv_from(double alpha, const HipVec& x, double beta, const HipVec& y0, HipVec& y) const {
    if (x.len() != ncols || y.len() != nrows || y0.len() != nrows || y0.nb() != y.nb()) throw LaError(DSH_E_INVALID, "gemv_from: shape mismatch");
    check(dsh_mat_gemv_from(ctx.raw(), nrows, ncols, nb, alpha, p, nb, x.ptr(), x.nb(), beta, y0.ptr(), y.ptr()), "gemv_from");
  }
};
struct HipMatViewMut {
  double* p = nullptr;
  int64_t nrows = 0, ncols = 0, nb = 1;
  HipContext ctx;
  // self = alpha*a*b + beta*self  (gemm_vo, matrix/cuda.rs:757-821); b may be a broadcast (nbatch 1) matrix
  void gemm_vo(double alpha, const HipMatView& a, const class HipMat& b, double beta);
};

// Owning dense matrix, column-major per system (CudaMat, matrix/cuda.rs:32-41)
class HipMat {
 public:
  HipMat() = default;
  static HipMat zeros(int64_t nrows, int64_t ncols, const HipContext& ctx) { HipMat m(nrows, ncols, ctx, true); m.set_band(0, 0); return m; }
  // BAND CONTAINER of a square matrix whose structure is declared (diffsol_hip.h dsh_mat_band_*): (kl + ku + 1) n entries per member instead of n^2, entry (i, j)
  // at ((j - i + kl) n + i) nbatch + b.  It takes what the Jacobian / mass / M - cJ containers of the implicit integrators need — scale_add_and_assign with
  // containers of the same bands, gemv, copy_from, the model's banded Jacobian evaluation, the LU factorisation — and refuses every other matrix operation.
  static HipMat zeros_banded(int64_t n, int kl, int ku, const HipContext& ctx) {
    if (kl < 0 || ku < 0 || kl + ku + 1 > n) throw LaError(DSH_E_INVALID, "zeros_banded: bad bandwidths");
    HipMat m(n, n, ctx, true, (int64_t)(kl + ku + 1) * n);
    m.packed_ = true; m.band_kl_ = kl; m.band_ku_ = ku;
    return m;
  }
  static HipMat from_diagonal_banded(const HipVec& v, int kl, int ku) {
    HipMat m = zeros_banded(v.len(), kl, ku, v.context());
    check(dsh_mat_band_from_diagonal(v.context().raw(), v.len(), v.nb(), kl, ku, v.ptr(), v.nb(), m.buf()), "from_diagonal_banded");
    return m;
  }
  bool packed() const { return packed_; }
  double* band_ptr() { if (!packed_) throw LaError(DSH_E_INVALID, "band_ptr: not a band container"); return buf(); }
  int64_t storage_entries() const { return packed_ ? (int64_t)(band_kl_ + band_ku_ + 1) * nrows_ : nrows_ * ncols_; }
  // data: batch-major, column-major per batch member ([b][col][row], matrix/cuda.rs:20-31)
  static HipMat from_vec(int64_t nrows, int64_t ncols, const std::vector<double>& data, const HipContext& ctx) {
    if ((int64_t)data.size() != nrows * ncols * ctx.nbatch()) throw LaError(DSH_E_INVALID, "from_vec: wrong data length");
    HipMat m(nrows, ncols, ctx, false);
    check(dsh_vec_upload(ctx.raw(), nrows * ncols, ctx.nbatch(), data.data(), m.ptr()), "HipMat::from_vec");
    return m;
  }
  static HipMat from_diagonal(const HipVec& v) {
    HipMat m(v.len(), v.len(), v.context(), true);  // large: storage deferred (see Buf)
    if (m.data_->p) {
      check(dsh_mat_from_diagonal(v.context().raw(), v.len(), v.nb(), v.ptr(), v.nb(), m.data_->p), "from_diagonal");
    } else {
      const HipVec keep = v.clone();  // the diagonal (n per member) is kept until the n x n storage is first touched
      m.data_->init = [keep](double* p) { check(dsh_mat_from_diagonal(keep.context().raw(), keep.len(), keep.nb(), keep.ptr(), keep.nb(), p), "from_diagonal"); };
    }
    m.set_band(0, 0);
    return m;
  }
  std::vector<double> clone_as_vec() const {
    dense_only("clone_as_vec");
    std::vector<double> out((size_t)(nrows_ * ncols_ * nb()));
    check(dsh_vec_download(ctx_.raw(), nrows_ * ncols_, nb(), ptr(), out.data()), "HipMat::clone_as_vec");
    return out;
  }
  int64_t nrows() const { return nrows_; }
  int64_t ncols() const { return ncols_; }
  int64_t nb() const { return ctx_.nbatch(); }
  const HipContext& context() const { return ctx_; }
  // Structure tag of the CONTENT: every entry outside |i - j| <= (kl below, ku above) is exactly zero.  Set by writers that know it (zeros, from_diagonal,
  // a model that declares the bandwidth of its Jacobian / mass matrix, the banded scale_add_and_assign); any other write access clears it.  It lets
  // M - cJ be assembled and factored on the band only (dsh_mat_scale_add_assign_banded, dsh_lu_factor_banded) — same arithmetic per entry.
  void set_band(int kl, int ku) { if (packed_) { if (kl > band_kl_ || ku > band_ku_) throw LaError(DSH_E_INVALID, "set_band: wider than the band container"); return; } band_kl_ = kl; band_ku_ = ku; }
  void clear_band() { if (!packed_) band_kl_ = band_ku_ = -1; }
  bool has_band() const { return band_kl_ >= 0 && band_ku_ >= 0; }
  int band_kl() const { return band_kl_; }
  int band_ku() const { return band_ku_; }
  double* ptr() { dense_only("ptr"); clear_band(); return buf(); }  // write access from outside: the content is no longer known to be banded
  const double* ptr() const { return buf(); }
  int64_t col_stride() const { return nrows_ * nb(); }

  HipVecView column(int64_t j) const { dense_only("column"); bounds(j); return HipVecView{ptr() + j * col_stride(), nrows_, nb(), ctx_}; }
  HipVecViewMut column_mut(int64_t j) { dense_only("column_mut"); bounds(j); return HipVecViewMut{ptr() + j * col_stride(), nrows_, nb(), ctx_}; }
  HipMatView columns(int64_t start, int64_t end) const { dense_only("columns"); if (start < 0 || end > ncols_ || start > end) throw LaError(DSH_E_INVALID, "columns: out of bounds"); return HipMatView{ptr() + start * col_stride(), nrows_, end - start, nb(), ctx_}; }
  HipMatViewMut columns_mut(int64_t start, int64_t end) { dense_only("columns_mut"); if (start < 0 || end > ncols_ || start > end) throw LaError(DSH_E_INVALID, "columns_mut: out of bounds"); return HipMatViewMut{ptr() + start * col_stride(), nrows_, end - start, nb(), ctx_}; }
  HipVec diagonal() const { dense_only("diagonal"); HipVec v = HipVec::zeros(nrows_, ctx_); check(dsh_mat_get_diagonal(ctx_.raw(), nrows_, nb(), ptr(), v.ptr()), "diagonal"); return v; }

  void copy_from(const HipMat& o) {
    same_shape(o);
    if (packed_ || o.packed_) {
      if (!(packed_ && o.packed_ && band_kl_ == o.band_kl_ && band_ku_ == o.band_ku_)) throw LaError(DSH_E_UNSUPPORTED, "copy_from: band containers of different structure");
      check(dsh_vec_copy(ctx_.raw(), storage_entries(), nb(), o.buf(), o.nb(), buf()), "HipMat::copy_from (band)");
      return;
    }
    check(dsh_vec_copy(ctx_.raw(), nrows_ * ncols_, nb(), o.ptr(), o.nb(), ptr()), "HipMat::copy_from");
  }
  void set_column(int64_t j, const HipVec& v) { dense_only("set_column"); if (v.len() != nrows_) throw LaError(DSH_E_INVALID, "set_column: length mismatch"); check(dsh_mat_set_column(ctx_.raw(), nrows_, ncols_, nb(), ptr(), j, v.ptr(), v.nb()), "set_column"); }
  // self = x + beta*y  (matrix/cuda.rs:1424-1458)
  void scale_add_and_assign(const HipMat& x, double beta, const HipMat& y) {
    same_shape(x); same_shape(y);
    if (packed_ || x.packed_ || y.packed_) {  // band containers of one structure: the same entry-wise x + beta*y over (kl + ku + 1) n entries
      if (!(packed_ && x.packed_ && y.packed_ && band_kl_ == x.band_kl_ && band_ku_ == x.band_ku_ && band_kl_ == y.band_kl_ && band_ku_ == y.band_ku_))
        throw LaError(DSH_E_UNSUPPORTED, "scale_add_and_assign: band containers of different structure");
      check(dsh_mat_scale_add_assign(ctx_.raw(), storage_entries(), nb(), buf(), x.buf(), x.nb(), beta, y.buf(), y.nb()), "scale_add_and_assign (band container)");
      return;
    }
    const bool tagged = x.has_band() && y.has_band() && nrows_ == ncols_;
    const int kl = tagged ? std::max(x.band_kl(), y.band_kl()) : -1, ku = tagged ? std::max(x.band_ku(), y.band_ku()) : -1;
    // banded assembly: the result's band must cover what this container holds now (else stale entries outside it would survive)
    if (tagged && has_band() && band_kl_ <= kl && band_ku_ <= ku && nrows_ >= 16 && (int64_t)(kl + ku + 1) * 2 <= nrows_) {
      check(dsh_mat_scale_add_assign_banded(ctx_.raw(), nrows_, nb(), kl, ku, buf(), x.ptr(), x.nb(), beta, y.ptr(), y.nb()), "scale_add_and_assign (banded)");
    } else {
      check(dsh_mat_scale_add_assign(ctx_.raw(), nrows_ * ncols_, nb(), buf(), x.ptr(), x.nb(), beta, y.ptr(), y.nb()), "scale_add_and_assign");
    }
    band_kl_ = kl; band_ku_ = ku;
  }
  // column i += alpha * column j  (matrix/cuda.rs:1048-1088)
  void column_axpy(double alpha, int64_t j, int64_t i) { dense_only("column_axpy"); bounds(i); bounds(j); check(dsh_mat_column_axpy(ctx_.raw(), nrows_, nb(), ptr(), alpha, j, i), "column_axpy"); }
  void gemv(double alpha, const HipVec& x, double beta, HipVec& y) const {
    if (packed_) {
      if (x.len() != ncols_ || y.len() != nrows_) throw LaError(DSH_E_INVALID, "gemv: shape mismatch");
      check(dsh_mat_band_gemv(ctx_.raw(), nrows_, nb(), band_kl_, band_ku_, alpha, buf(), x.ptr(), x.nb(), beta, y.ptr()), "gemv (band container)");
      return;
    }
    columns(0, ncols_).gemv_o(alpha, x, beta, y);
  }
  void gemm(double alpha, const HipMat& a, const HipMat& b, double beta) {
    dense_only("gemm"); a.dense_only("gemm"); b.dense_only("gemm");
    if (a.nrows_ != nrows_ || b.ncols_ != ncols_ || a.ncols_ != b.nrows_) throw LaError(DSH_E_INVALID, "gemm: shape mismatch");
    check(dsh_mat_gemm(ctx_.raw(), nrows_, ncols_, a.ncols_, nb(), alpha, a.ptr(), a.nb(), b.ptr(), b.nb(), beta, ptr()), "gemm");
  }
  HipMat mat_mul(const HipMat& b) const { HipMat r = HipMat::zeros(nrows_, b.ncols_, ctx_); r.gemm(1.0, *this, b, 0.0); return r; }
  // resize_cols preserving data (matrix/cuda.rs resize_cols; used by OdeSolverMethod::solve, method.rs:1000-1003)
  void resize_cols(int64_t ncols) {
    dense_only("resize_cols");
    if (ncols == ncols_) return;
    HipMat m(nrows_, ncols, ctx_, true);
    int64_t keep = ncols < ncols_ ? ncols : ncols_;
    if (keep > 0) check(dsh_d2d(ctx_.raw(), m.ptr(), ptr(), (int64_t)sizeof(double) * keep * col_stride()), "resize_cols");
    ctx_.sync();
    *this = m;
  }
  void swap(HipMat& o) { std::swap(*this, o); }

 private:
  int band_kl_ = -1, band_ku_ = -1;
  bool packed_ = false;
  void dense_only(const char* what) const { if (packed_) throw LaError(DSH_E_UNSUPPORTED, std::string(what) + ": not available on a band container"); }
  // The storage of a LARGE zero matrix (>= 256 MB) is allocated when it is first touched: a solver object owns the n x n containers of the host-driven path
  // (Jacobian, mass, M - cJ: 2 MB per member each at n = 512) whether or not a solve ever uses them, and an ensemble integrated by the device-resident
  // kernels never does — 32 768 members of a 512-state model would not fit the device otherwise.  Copies share the buffer, allocated or not.
  struct Buf {
    HipContext ctx; int64_t bytes; double* p = nullptr;
    Buf(const HipContext& c, int64_t b) : ctx(c), bytes(b) {}
    ~Buf() { if (p) dsh_free(ctx.raw(), p); }
    std::function<void(double*)> init;  // content of a deferred matrix that is not all zeros (from_diagonal): written when the storage is made
    double* get(bool zero) {
      if (!p) {
        void* d = nullptr;
        check(dsh_malloc(ctx.raw(), bytes, (zero && !init) ? 1 : 0, &d), "HipMat alloc");
        p = (double*)d;
        if (init) { init(p); init = nullptr; }
      }
      return p;
    }
  };
  HipMat(int64_t nrows, int64_t ncols, const HipContext& ctx, bool zero, int64_t packed_entries = 0) : nrows_(nrows), ncols_(ncols), ctx_(ctx) {
    const int64_t bytes = (int64_t)sizeof(double) * (packed_entries > 0 ? packed_entries : nrows * ncols) * ctx.nbatch();
    data_ = std::make_shared<Buf>(ctx, bytes);
    if (!(zero && bytes >= ((int64_t)256 << 20))) (void)data_->get(zero);
  }
  double* buf() const { return data_ ? data_->get(true) : nullptr; }
  void bounds(int64_t j) const { if (j < 0 || j >= ncols_) throw LaError(DSH_E_INVALID, "Column index out of bounds"); }
  void same_shape(const HipMat& o) const { if (o.nrows_ != nrows_ || o.ncols_ != ncols_) throw LaError(DSH_E_INVALID, "Matrix shape mismatch"); }
  int64_t nrows_ = 0, ncols_ = 0;
  HipContext ctx_;
  std::shared_ptr<Buf> data_;
};

inline void HipMatViewMut::gemm_vo(double alpha, const HipMatView& a, const HipMat& b, double beta) {
  if (a.nrows != nrows || b.ncols() != ncols || a.ncols != b.nrows()) throw LaError(DSH_E_INVALID, "gemm_vo: shape mismatch");
  check(dsh_mat_gemm(ctx.raw(), nrows, ncols, a.ncols, nb, alpha, a.p, a.nb, b.ptr(), b.nb(), beta, p), "gemm_vo");
}

// LinearOp boundary used by LinearSolver::{set_sparsity,set_linearisation} (diffsol-la/src/linear_op.rs:13-37)
struct LinearOpRef {
  virtual ~LinearOpRef() = default;
  virtual int64_t nrows() const = 0;
  virtual int64_t ncols() const = 0;
  virtual const HipContext& context() const = 0;
  virtual void matrix_inplace(HipMat& y) const = 0;
  // the operator's matrices live in band containers of these bandwidths (HipMat::zeros_banded): the solver then keeps its own matrix and its factors banded too
  virtual bool packed_band(int* kl, int* ku) const { (void)kl; (void)ku; return false; }
};

// HipLU: LinearSolver<HipMat> (CudaLU, linear_solver/cuda/lu.rs:15-191)
class HipLU {
 public:
  HipLU() = default;
  void set_sparsity(const LinearOpRef& op) {  // lu.rs:148-190
    if (op.nrows() != op.ncols()) throw LaError(DSH_E_INVALID, "LinearSolverMatrixNotSquare");
    ctx_ = op.context();
    int kl = 0, ku = 0;
    dsh_lu* lu = nullptr;
    if (op.packed_band(&kl, &ku)) {
      matrix_ = HipMat::zeros_banded(op.nrows(), kl, ku, ctx_);
      check(dsh_lu_create_banded(ctx_.raw(), op.nrows(), ctx_.nbatch(), std::max(1, std::max(kl, ku)), &lu), "HipLU::set_sparsity (band)");
    } else {
      matrix_ = HipMat::zeros(op.nrows(), op.ncols(), ctx_);
      check(dsh_lu_create(ctx_.raw(), op.nrows(), ctx_.nbatch(), &lu), "HipLU::set_sparsity");
    }
    HipContext keep = ctx_;
    lu_ = std::shared_ptr<dsh_lu>(lu, [keep](dsh_lu* p) { dsh_lu_destroy(p); });
    factored_ = false;
  }
  void set_linearisation(const LinearOpRef& op) {  // lu.rs:59-97
    if (!lu_) throw LaError(DSH_E_NOT_SETUP, "LinearSolverNotSetup");
    op.matrix_inplace(matrix_);
    const HipMat& m = matrix_;  // const access keeps the structure tag
    if (m.packed()) check(dsh_lu_factor_packed(lu_.get(), m.ptr(), m.band_kl(), m.band_ku()), "HipLU::set_linearisation (band container)");
    else if (m.has_band()) check(dsh_lu_factor_banded(lu_.get(), m.ptr(), m.band_kl(), m.band_ku()), "HipLU::set_linearisation (declared band)");
    else check(dsh_lu_factor(lu_.get(), m.ptr()), "HipLU::set_linearisation");
    factored_ = true;
  }
  // returns false on LuSolveFailed (zero pivot), throws LuNotInitialized
  bool solve_in_place(HipVec& x) const {  // lu.rs:99-146
    if (!lu_ || !factored_) throw LaError(DSH_E_NOT_SETUP, "LuNotInitialized");
    if (x.len() != matrix_.nrows()) throw LaError(DSH_E_INVALID, "LinearSolverMatrixVectorNotCompatible");
    int rc = dsh_lu_solve(lu_.get(), x.ptr());
    if (rc == DSH_E_SINGULAR) return false;
    check(rc, "HipLU::solve_in_place");
    return true;
  }
  // solve_in_place(x) and x.squared_norm(y, atol, rtol) with one wait for both results; false on LuSolveFailed
  bool solve_in_place_and_norm(HipVec& x, const HipVec& y, const HipVec& atol, double rtol, double* norm) const {
    if (!lu_ || !factored_) throw LaError(DSH_E_NOT_SETUP, "LuNotInitialized");
    if (x.len() != matrix_.nrows()) throw LaError(DSH_E_INVALID, "LinearSolverMatrixVectorNotCompatible");
    int rc = dsh_lu_solve_squared_norm(lu_.get(), x.ptr(), y.ptr(), y.nb(), atol.ptr(), atol.nb(), rtol, norm);
    if (rc == DSH_E_SINGULAR) return false;
    check(rc, "HipLU::solve_in_place_and_norm");
    return true;
  }
  dsh_lu* raw() const { return lu_.get(); }
  bool is_setup() const { return (bool)lu_; }
  void mark_factored() { factored_ = true; }
  HipMat& matrix() { return matrix_; }

 private:
  HipContext ctx_;
  HipMat matrix_;
  std::shared_ptr<dsh_lu> lu_;
  bool factored_ = false;
};

}  // namespace diffsol_hip
