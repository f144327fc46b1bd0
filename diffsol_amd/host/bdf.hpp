// Host-side BDF integrator over the HIP backend: mirror of crates/diffsol/src/ode_solver/bdf.rs (Bdf), op/bdf.rs (BdfCallable) and
// ode_solver/bdf_state.rs (BdfState).  Variable-order (1-5) NDF with the fixed-leading-coefficient difference array D (n x 8),
// lock-step over the ensemble: one t / h / order for all systems, scalar decisions taken on max-over-batch norms
// (diffsol-la/src/vector/cuda.rs:100-115).  Two execution modes, bit-identical by construction:
//   - trait mode : every step of the algorithm is the reference's sequence of Vector/Matrix/LinearSolver calls (1 dsh_* call each);
//   - fused mode : prepare / Newton iteration / Jacobian refresh / accept are each ONE device launch (dsh_bdf_*, dsh_jac_factor),
//                  3-4 launches and 2-3 host syncs per accepted step instead of ~40 launches.
//   bdf.rs:244-368 _new, :433-463 _compute_r, :465-506 _jacobian_updates, :508-577 _update_step_size, :646-692 _update_diff/_predict,
//   :694-731 handle_tstop, :767-811 interpolation, :812-932 error control, :1277-1589 step
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "ode.hpp"

namespace diffsol_hip {

// Optional host-side wall-clock accounting of the fused step loop (DSH_PROFILE_HOST=1 prints it when the solver is destroyed).
struct HostProfile {
  bool on = false;
  double t_launch_newton = 0, t_wait_newton = 0, t_launch_accept = 0, t_wait_accept = 0, t_jac = 0, t_prepare = 0;
  long n_launch_newton = 0, n_wait_newton = 0, n_launch_accept = 0, n_wait_accept = 0, n_jac = 0, n_prepare = 0;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  ~HostProfile() {
    if (!on) return;
    std::fprintf(stderr, "[host profile] launch_newton %ld x %.2f us | wait_newton %ld x %.2f us | launch_accept %ld x %.2f us | wait_accept %ld x %.2f us | jac %ld x %.2f us | prepare %ld x %.2f us\n",
                 n_launch_newton, 1e6 * t_launch_newton / std::max(1L, n_launch_newton), n_wait_newton, 1e6 * t_wait_newton / std::max(1L, n_wait_newton),
                 n_launch_accept, 1e6 * t_launch_accept / std::max(1L, n_launch_accept), n_wait_accept, 1e6 * t_wait_accept / std::max(1L, n_wait_accept),
                 n_jac, 1e6 * t_jac / std::max(1L, n_jac), n_prepare, 1e6 * t_prepare / std::max(1L, n_prepare));
  }
};
#define DSH_PROF(field, stmt)                                  \
  do {                                                         \
    if (prof_.on) {                                            \
      double _t0 = HostProfile::now();                         \
      stmt;                                                    \
      prof_.t_##field += HostProfile::now() - _t0;             \
      prof_.n_##field += 1;                                    \
    } else {                                                   \
      stmt;                                                    \
    }                                                          \
  } while (0)

// op/bdf.rs:15-300
class BdfCallable : public NonLinearOpRef {
 public:
  explicit BdfCallable(const OdeEquations& eqn)
      : eqn_(eqn), psi_neg_y0_(HipVec::zeros(eqn.nstates(), eqn.context())), tmp_(HipVec::zeros(eqn.nstates(), eqn.context())),
        rhs_jac_(HipMat::zeros(eqn.nstates(), eqn.nstates(), eqn.context())) {
    const int64_t n = eqn.nstates();
    int pkl = 0, pku = 0;
    if (eqn.packed_band(&pkl, &pku)) {  // declared narrow band, identity mass: band containers for f_y, M and (through packed_band below) M - cJ and its factors
      packed_ = true; pkl_ = pkl; pku_ = pku;
      rhs_jac_ = HipMat::zeros_banded(n, pkl, pku, eqn.context());
      mass_jac_ = HipMat::from_diagonal_banded(HipVec::from_element(n, 1.0, eqn.context()), pkl, pku);
    } else
    if (!eqn.has_mass()) mass_jac_ = HipMat::from_diagonal(HipVec::from_element(n, 1.0, eqn.context()));  // :138-141
    else mass_jac_ = HipMat::zeros(n, n, eqn.context());
  }
  int64_t nstates() const override { return eqn_.nstates(); }
  const HipContext& context() const override { return eqn_.context(); }
  bool packed_band(int* kl, int* ku) const override { if (!packed_) return false; *kl = pkl_; *ku = pku_; return true; }
  void set_c(double h, double alpha) { c_ = h * alpha; }
  void set_c_value(double c) { c_ = c; }
  double c() const { return c_; }
  void set_psi(const HipMat& diff, const std::vector<double>& gamma, const std::vector<double>& alpha, int order, HipVec& psi) const {  // :182-196
    psi.axpy_v(gamma[1], diff.column(1), 0.0);
    for (int i = 2; i <= order; ++i) psi.axpy_v(gamma[(size_t)i], diff.column(i), 1.0);
    psi.mul_assign(scale(alpha[(size_t)order]));
  }
  void set_psi_and_y0(const HipMat& diff, const std::vector<double>& gamma, const std::vector<double>& alpha, int order, const HipVec& y0) {  // :197-210
    set_psi(diff, gamma, alpha, order, psi_neg_y0_);
    psi_neg_y0_.sub_assign(y0);
  }
  void set_jacobian_is_stale() { jacobian_is_stale_ = true; }
  bool jacobian_is_stale() const { return jacobian_is_stale_; }
  void clear_jacobian_is_stale() { jacobian_is_stale_ = false; }
  // F(y) = M (y - y0 + psi) - c f(y)   :240-256
  void call_inplace(const HipVec& x, double t, HipVec& y) override {
    eqn_.rhs_call_inplace(x, t, y);
    tmp_.copy_from(x);
    tmp_.add_assign(psi_neg_y0_);
    if (eqn_.has_mass()) eqn_.mass_gemv_inplace(tmp_, t, -c_, y);
    else y.axpy(1.0, tmp_, -c_);
  }
  // M - c f'(y)   :273-300
  void jacobian_inplace(const HipVec& x, double t, HipMat& y) override {
    if (jacobian_is_stale_) {
      eqn_.rhs_jacobian_inplace(x, t, rhs_jac_);
      if (eqn_.has_mass()) eqn_.mass_matrix_inplace(t, mass_jac_);
      y.scale_add_and_assign(mass_jac_, -c_, rhs_jac_);
      jacobian_is_stale_ = false;
    } else {
      y.scale_add_and_assign(mass_jac_, -c_, rhs_jac_);
    }
  }
  HipVec& psi_neg_y0() { return psi_neg_y0_; }
  HipMat& rhs_jac() { return rhs_jac_; }
  HipMat& mass_jac() { return mass_jac_; }

 private:
  const OdeEquations& eqn_;
  HipVec psi_neg_y0_, tmp_;
  double c_ = 0.0;
  HipMat rhs_jac_, mass_jac_;
  bool packed_ = false;
  int pkl_ = 0, pku_ = 0;
  bool jacobian_is_stale_ = true;
};

class Bdf : public OdeSolverMethod {
 public:
  static constexpr int MAX_ORDER = 5;  // bdf_state.rs:44

  // problem.bdf::<LS>() (problem.rs:597-655): consistent state with solver_order = 1, then Bdf::new
  explicit Bdf(const OdeSolverProblem& problem)
      : pr_(problem), convergence_(problem.rtol, &problem.atol, problem.ode_options.nonlinear_solver_tolerance), op_(*problem.eqn),
        jacobian_update_(problem.ode_options) {
    const OdeSolverOptions& o = problem.ode_options;
    minimum_timestep_ = o.min_timestep;
    maximum_error_test_failures_ = o.max_error_test_failures;
    maximum_newton_fails_ = o.max_nonlinear_solver_failures;
    maximum_timestep_growth_ = o.max_timestep_growth.value_or(2.0);
    minimum_timestep_growth_ = o.min_timestep_growth.value_or(2.0);
    maximum_timestep_shrink_ = o.max_timestep_shrink.value_or(0.9);
    minimum_timestep_shrink_ = o.min_timestep_shrink.value_or(0.5);
    // forward sensitivities run on the trait operations (the fused step kernels integrate the state equations only)
    fused_ = problem.use_fused_kernels && !problem.sens && problem.eqn->fused_model(&model_, &model_size_);
    // Run-time-sized models have no fused Newton kernel, but the difference-array kernels (rescale, predict, accept + order-selection norms) are model
    // independent: with enough members to fill the chip with one lane per system they replace ~20 vector launches per step (same bits, tested against
    // the trait composition).  DSH_FUSE_LA=0 keeps the 1:1 trait operations.
    {
      const char* e = std::getenv("DSH_FUSE_LA");
      fused_la_ = problem.use_fused_kernels && !problem.sens && !fused_ && problem.context().nbatch() >= 8192 && !(e && e[0] == '0');
    }

    StateCommon sc = new_and_consistent(problem, 1);
    y_ = sc.y; dy_ = sc.dy; t_ = sc.t; h_ = sc.h;
    if (problem.sens) {
      // bdf_state_sens -> new_with_sensitivities_and_consistent (state.rs:1032-1083): s_j = SensInit(t0) (:1157-1175), ds_j = SensRhs(s_j) about (y0, t0)
      // (set_consistent_augmented :167-186); with a singular mass matrix InitOp on the sensitivity equations (:187-238), below
      const int64_t n0 = problem.eqn->nstates(), npar = problem.eqn->nparams();
      const HipContext& c0 = problem.context();
      sens_mat_ = HipMat::zeros(n0, npar, c0);
      sens_y_ = HipVec::zeros(n0, c0);
      HipMat s0 = HipMat::zeros(n0, npar, c0);
      problem.eqn->init_sens_inplace(t_, s0);
      sens_update_state(y_, t_);
      for (int64_t j = 0; j < npar; ++j) {
        HipVec sj = HipVec::zeros(n0, c0), dsj = HipVec::zeros(n0, c0);
        sj.copy_from_view(s0.column(j));
        sens_rhs_call((int)j, sj, t_, dsj);
        s_.push_back(std::move(sj));
        ds_.push_back(std::move(dsj));
      }
      if (problem.eqn->has_mass()) sens_set_consistent();
    }

    // kappa table and derived constants (bdf.rs:253-276)
    const double kappa[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    alpha_ = {0.0}; gamma_ = {0.0}; error_const2_ = {1.0};
    for (int i = 1; i <= MAX_ORDER; ++i) {
      double i_t = (double)i, one_over_i = 1.0 / i_t, one_over_i_plus_one = 1.0 / (i_t + 1.0);
      gamma_.push_back(gamma_[(size_t)i - 1] + one_over_i);
      alpha_.push_back(1.0 / ((1.0 - kappa[i]) * gamma_[(size_t)i]));
      double e = kappa[i] * gamma_[(size_t)i] + one_over_i_plus_one;
      error_const2_.push_back(e * e);
    }
    convergence_.set_max_iter(o.max_nonlinear_solver_iterations);

    const int64_t n = problem.eqn->nstates();
    const HipContext& ctx = problem.context();
    op_.set_c(h_, alpha_[(size_t)order_]);
    nonlinear_solver_.set_problem(op_);
    // first Jacobian + LU (bdf.rs:289-293).  State and time cannot change between here and the first step, so the work is done at the top of that step
    // (ensure_linearised) with the same operands — INCLUDING the c of this moment: set_stop_time may shorten the first step before it is taken
    // (handle_tstop -> update_step_size changes op.c), and the reference then keeps the factors of M - c0 J it made here unless its update rule asks
    // for new ones.  An ensemble that is handed to the device-resident kernels never touches the n x n containers of this path (lazily allocated,
    // hip_la.hpp) — at n = 512 they would not fit the device from ~30 000 members on.
    first_linearisation_pending_ = true;
    first_linearisation_c_ = op_.c();
    diff_ = HipMat::zeros(n, MAX_ORDER + 3, ctx);
    initialise_diff_to_first_order();
    if (problem.eqn->nroots() > 0) { root_finder_.emplace(problem.eqn->nroots(), n, ctx); root_finder_->init(*problem.eqn, y_, t_); }
    diff_tmp_ = HipMat::zeros(n, MAX_ORDER + 3, ctx);
    y_delta_ = HipVec::zeros(n, ctx);
    if (const char* env = std::getenv("DSH_PROFILE_HOST")) prof_.on = std::string(env) == "1";
    if (const char* env = std::getenv("DSH_NEWTON_NIT")) nit_ = std::max(1, std::min(4, std::atoi(env)));
    if (fused_) { ybuf_[0] = HipVec::zeros(n * nit_, ctx); ybuf_[1] = HipVec::zeros(n * nit_, ctx); }
    y_new_ = y_delta_.ptr();
    y_predict_ = HipVec::zeros(n, ctx);
    if (const char* env = std::getenv("DSH_NEWTON_PIPELINE")) pipeline_ = std::string(env) != "0";
    if (const char* env = std::getenv("DSH_FUSE_ACCEPT")) fuse_accept_ = std::string(env) != "0";
    d_tmp_ = HipVec::zeros(n, ctx);
    u_ = compute_r(order_, 1.0);
    statistics_.number_of_linear_solver_setups = 1;
    statistics_.number_of_linear_solver_setups_from_checkpoint = 1;
    if (problem.sens) {  // new_augmented (bdf.rs:384-432): initialise_sdiff_to_first_order (bdf_state.rs:80-91); s_op = BdfCallable::new_no_jacobian (c = 0!)
      for (size_t j = 0; j < s_.size(); ++j) {
        HipMat sd = HipMat::zeros(n, MAX_ORDER + 3, ctx);
        sd.column_mut(0).copy_from(s_[j]);
        sd.column_mut(1).copy_from(ds_[j]);
        sd.column_mut(1).mul_assign(scale(h_));
        sdiff_.push_back(std::move(sd));
        s_deltas_.push_back(HipVec::zeros(n, ctx));
      }
      s_predict_ = HipVec::zeros(n, ctx);
      s_psi_neg_y0_ = HipVec::zeros(n, ctx);
      s_tmp_ = HipVec::zeros(n, ctx);
    }
  }

  // ---- forward sensitivities (bdf.rs:934-989 sensitivity_solve; SensRhs ode_equations/sens_equations.rs:72-180)
  // SensRhs::update_state: df/dp and the linearisation point
  void sens_update_state(const HipVec& y, double t) {
    pr_.eqn->rhs_sens_inplace(y, t, sens_mat_);
    sens_y_.copy_from(y);
  }
  void sens_set_consistent() {
    sens_set_consistent_augmented(pr_, t_, sens_y_, s_, ds_, [this](int j, const HipVec& x, double t, HipVec& y) { sens_rhs_call(j, x, t, y); });
  }
  // SensRhs::call_inplace: J(sens_y) x + (df/dp)[:, index]
  void sens_rhs_call(int index, const HipVec& x, double t, HipVec& y) const {
    pr_.eqn->rhs_jac_mul_inplace(sens_y_, t, x, y);
    y.add_assign(sens_mat_.column(index));  // Matrix::add_column_to_vector
  }
  // the BdfCallable of the sensitivity equations (op/bdf.rs:240-256) for parameter `index`: F(s) = M (s - s0 + psi) - c SensRhs(s).
  // Its c is set by _update_step_size only (bdf.rs:551-553): new_augmented never calls set_c on it, so it is 0 until the first step-size change —
  // the reference's behaviour, kept because the reference's step counts only reproduce with it (tests/test_oracle_golden.py).
  struct SensOp : NonLinearOpRef {
    Bdf& b; int index;
    SensOp(Bdf& b_, int i) : b(b_), index(i) {}
    int64_t nstates() const override { return b.n(); }
    const HipContext& context() const override { return b.ctx(); }
    void call_inplace(const HipVec& x, double t, HipVec& y) override {
      b.sens_rhs_call(index, x, t, y);
      b.s_tmp_.copy_from(x);
      b.s_tmp_.add_assign(b.s_psi_neg_y0_);
      if (b.pr_.eqn->has_mass()) b.pr_.eqn->mass_gemv_inplace(b.s_tmp_, t, -b.s_c_, y);  // SensEquations::mass is the state equations' mass (sens_equations.rs:305-307)
      else y.axpy(1.0, b.s_tmp_, -b.s_c_);
    }
    void jacobian_inplace(const HipVec&, double, HipMat&) override { throw LaError(DSH_E_UNSUPPORTED, "the sensitivity operator shares the state equations' factors"); }
  };
  // one Newton solve per parameter with the factors of the state equations; false = SensitivitySolveFailed
  bool sensitivity_solve(double t_new) {
    const int order = order_;
    sens_update_state(y_predict_, t_new);  // `y_new = &self.y_predict` (bdf.rs:941)
    for (size_t j = 0; j < sdiff_.size(); ++j) {
      s_predict_.fill(0.0);  // _predict_using_diff
      for (int i = 0; i <= order; ++i) s_predict_.add_assign(sdiff_[j].column(i));
      s_psi_neg_y0_.axpy_v(gamma_[1], sdiff_[j].column(1), 0.0);  // set_psi_and_y0
      for (int i = 2; i <= order; ++i) s_psi_neg_y0_.axpy_v(gamma_[(size_t)i], sdiff_[j].column(i), 1.0);
      s_psi_neg_y0_.mul_assign(scale(alpha_[(size_t)order]));
      s_psi_neg_y0_.sub_assign(s_predict_);
      s_[j].copy_from(s_predict_);
      SensOp sop(*this, (int)j);
      if (nonlinear_solver_.solve_in_place(sop, s_[j], t_new, s_predict_, convergence_, line_search_) != NlError::Ok) return false;  // `?` before the count
      statistics_.number_of_nonlinear_solver_iterations += convergence_.niter();
      s_deltas_[j].copy_from(s_[j]);
      s_deltas_[j].sub_assign(s_predict_);
    }
    return true;
  }
  // OdeSolverMethod::interpolate_sens (bdf.rs:1162-1215)
  void interpolate_sens_inplace(double t, std::vector<HipVec>& out) const {
    const bool is_forward = h_ > 0.0;
    if ((is_forward && t > t_) || (!is_forward && t < t_)) throw DSH_ODE_ERR(InterpolationTimeAfterCurrentTime);
    out.clear();
    for (size_t j = 0; j < sdiff_.size(); ++j) {
      HipVec v = HipVec::zeros(n(), ctx());
      double time_factor = 1.0;
      v.copy_from_view(sdiff_[j].column(0));
      for (int i = 0; i < order_; ++i) {
        double i_t = (double)i;
        time_factor *= (t - (t_ - h_ * i_t)) / (h_ * (1.0 + i_t));
        v.axpy_v(time_factor, sdiff_[j].column(i + 1), 1.0);
      }
      out.push_back(std::move(v));
    }
  }
  const std::vector<HipVec>& sens() const { return s_; }

  // _compute_r (bdf.rs:433-463), column-major (order+1)^2, kept on the host (the reference keeps it in an nbatch = 1 context)
  static std::vector<double> compute_r(int order, double factor) {
    const int nrows = order + 1, ncols = order + 1;
    std::vector<double> r((size_t)(nrows * ncols), 0.0);
    for (int j = 0; j < ncols; ++j) r[(size_t)(j * nrows)] = 1.0;
    for (int j = 1; j < ncols; ++j)
      for (int i = 1; i < nrows; ++i) {
        size_t idx = (size_t)(j * nrows + i);
        r[idx] = r[idx - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
      }
    return r;
  }
  // R*U with the reference's small-matrix gemm accumulation order (nalgebra blas.rs gemm -> gemv -> axcpy)
  static std::vector<double> mat_mul_small(const std::vector<double>& a, const std::vector<double>& b, int n) {
    std::vector<double> out((size_t)(n * n));
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        double acc = a[(size_t)(0 * n + i)] * b[(size_t)(j * n + 0)];
        for (int k = 1; k < n; ++k) acc = a[(size_t)(k * n + i)] * b[(size_t)(j * n + k)] + acc;
        out[(size_t)(j * n + i)] = acc;
      }
    return out;
  }

  void ensure_linearised() {
    if (!first_linearisation_pending_) return;
    const double c_now = op_.c();
    op_.set_c_value(first_linearisation_c_);  // the constructor's factorisation, made late
    reset_jacobian();
    op_.set_c_value(c_now);
  }
  OdeSolverStopReason step() override {  // bdf.rs:1277-1589
    ensure_linearised();
    double safety = 0.0, error_norm = 0.0;
    const long old_num_error_test_failures = statistics_.number_of_error_test_failures;
    bool convergence_fail = false;
    if (is_state_modified_) {  // bdf.rs:1290-1318: the state was moved (state_mut_back): restart from first order
      if (root_finder_) root_finder_->init(*pr_.eqn, y_, t_);
      n_equal_steps_ = 0;
      prediction_valid_ = false;
      prelaunch_valid_ = false;
      initialise_diff_to_first_order();
      u_ = compute_r(1, 1.0);
      is_state_modified_ = false;
      const double c = h_ * alpha_[(size_t)order_];
      op_.set_c(h_, alpha_[(size_t)order_]);
      jacobian_updates(c, SolverState::StepSuccess);
      prev_error_norm_.reset();
      if (tstop_) set_stop_time(*tstop_);
    }
    predict_forward();
    while (true) {
      const int order = order_;
      if (!fused_) y_delta_.copy_from(y_predict_);  // fused mode: the first Newton launch reads the predictor directly
      double fused_err_sq = 0.0;
      NlError solve_result = fused_ ? newton_fused(fused_err_sq)
                                    : nonlinear_solver_.solve_in_place(op_, y_delta_, t_predict_, y_predict_, convergence_, line_search_);
      statistics_.number_of_nonlinear_solver_iterations += convergence_.niter();
      if (solve_result == NlError::Ok && fused_la_) { d_tmp_.copy_from(y_delta_); d_tmp_.sub_assign(y_predict_); }  // y_delta_ keeps y_new for the accept kernel
      else if (solve_result == NlError::Ok && !fused_) y_delta_.sub_assign(y_predict_);  // fused mode keeps y_new; d is formed in-kernel
      if (solve_result == NlError::Ok && pr_.sens && !sensitivity_solve(t_predict_)) solve_result = NlError::NewtonDiverged;  // SensitivitySolveFailed (bdf.rs:1355-1360)
      if (solve_result != NlError::Ok) {
        statistics_.number_of_nonlinear_solver_fails += 1;
        if (statistics_.number_of_nonlinear_solver_fails > maximum_newton_fails_) throw DSH_ODE_ERR(TooManyNonlinearSolverFailures);
        if (convergence_fail) {
          prev_error_norm_.reset();
          double new_h = update_step_size(0.3);
          jacobian_updates(new_h * alpha_[(size_t)order], SolverState::SecondConvergenceFail);
          predict_forward();
        } else {
          prev_error_norm_.reset();
          jacobian_updates(h_ * alpha_[(size_t)order], SolverState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      // error_control (bdf.rs:826-835): squared norm of d weighted by the OLD state, times error_const2[order-1]
      double err_sq = fused_ ? fused_err_sq : (fused_la_ ? d_tmp_.squared_norm(y_, pr_.atol, pr_.rtol) : y_delta_.squared_norm(y_, pr_.atol, pr_.rtol));
      error_norm = std::fmax(0.0, err_sq * error_const2_[(size_t)order_ - 1]);
      if (pr_.sens && pr_.sens_error_control)  // bdf.rs:844-858 — error_const2[order], not [order - 1]
        for (size_t j = 0; j < sdiff_.size(); ++j)
          error_norm = std::fmax(error_norm, s_deltas_[j].squared_norm(s_[j], pr_.sens_atol, pr_.sens_rtol) * error_const2_[(size_t)order_]);
      double maxiter = (double)convergence_.max_iter(), niter = (double)convergence_.niter();
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + niter);
      if (error_norm <= 1.0) break;
      double factor = safety * pi_controller_raw(error_norm, prev_error_norm_, pr_.ode_options.pi_control_integral, pr_.ode_options.pi_control_proportional, order + 1);
      prev_error_norm_.reset();
      if (factor < minimum_timestep_shrink_) factor = minimum_timestep_shrink_;
      double new_h = update_step_size(factor);
      jacobian_updates(new_h * alpha_[(size_t)order], SolverState::ErrorTestFail);
      predict_forward();
      statistics_.number_of_error_test_failures += 1;
      if (statistics_.number_of_error_test_failures - old_num_error_test_failures >= maximum_error_test_failures_) throw DSH_ODE_ERR(TooManyErrorTestFailures);
    }

    // take the accepted step (bdf.rs:1469-1478); order-selection norms are produced by the same launch in fused mode
    const bool will_select_order = (n_equal_steps_ + 1) > order_;
    double sel_norms[2] = {0.0, 0.0};
    if (fused_) {
      int64_t accept_ticket = 0;
      if (pipeline_ && fuse_accept_) {
        // accept + the first Newton launch of the NEXT step in one launch (see the comment below): state, prediction and psi stay in registers.
        // y_new_ may live in group buffer 0, which the Newton part overwrites: every lane reads its y_new before it stores an iterate, and a
        // lane only touches its own member, so the aliasing is harmless.
        prelaunch_nit_ = std::min(nit_, convergence_.max_iter());
        DSH_PROF(launch_accept, check(dsh_bdf_accept_newton_async(ctx().raw(), model_, model_size_, nb(), order_, h_, diff_.ptr(), y_predict_.ptr(), y_new_, y_.ptr(),
                                        dy_.ptr(), pr_.atol.ptr(), pr_.atol.nb(), pr_.rtol, gamma_.data(), alpha_[(size_t)order_], op_.psi_neg_y0().ptr(),
                                        t_predict_ + h_, op_.c(), prelaunch_nit_, ybuf_[0].ptr(), pr_.eqn->params().ptr(),
                                        nonlinear_solver_.linear_solver().raw(), &accept_ticket, &prelaunch_ticket_),
              "dsh_bdf_accept_newton_async"));
        t_ = t_predict_;
        prediction_valid_ = true;
        prelaunch_valid_ = true;
      } else {
      DSH_PROF(launch_accept, check(dsh_bdf_accept_step_async(ctx().raw(), n(), nb(), order_, h_, diff_.ptr(), y_predict_.ptr(), y_new_, y_.ptr(), dy_.ptr(), pr_.atol.ptr(),
                                      pr_.atol.nb(), pr_.rtol, gamma_.data(), alpha_[(size_t)order_], op_.psi_neg_y0().ptr(), &accept_ticket),
            "dsh_bdf_accept_step_async"));
      t_ = t_predict_;
      prediction_valid_ = true;  // y_predict / psi now hold the next step's prediction for (order_, h_)
      }
      if (pipeline_ && !fuse_accept_) {
        // Speculatively enqueue the first Newton launch of the NEXT step (valid if the controller leaves h, the order and the LU
        // factors alone — the common case) before waiting for this step's order-selection norms: the GPU stays busy across the step.
        // The accept launch above reads y_new_ (possibly in group buffer 0) and is stream-ordered before this launch overwrites it.
        prelaunch_nit_ = std::min(nit_, convergence_.max_iter());
        launch_newton(prelaunch_nit_, y_predict_.ptr(), 0, t_ + h_, &prelaunch_ticket_);
        prelaunch_valid_ = true;
      }
      if (will_select_order) {
        double r[3];
        DSH_PROF(wait_accept, check(dsh_reduction_wait(ctx().raw(), accept_ticket, r), "dsh_reduction_wait(accept)"));
        sel_norms[0] = r[0];
        sel_norms[1] = r[1];
      }
    } else if (fused_la_) {
      int64_t accept_ticket = 0;
      check(dsh_bdf_accept_step_async(ctx().raw(), n(), nb(), order_, h_, diff_.ptr(), y_predict_.ptr(), y_delta_.ptr(), y_.ptr(), dy_.ptr(), pr_.atol.ptr(), pr_.atol.nb(),
                                      pr_.rtol, gamma_.data(), alpha_[(size_t)order_], op_.psi_neg_y0().ptr(), &accept_ticket),
            "dsh_bdf_accept_step_async");
      t_ = t_predict_;
      prediction_valid_ = true;  // y_predict / psi now hold the next step's prediction for (order_, h_)
      if (will_select_order) {
        double r[3];
        check(dsh_reduction_wait(ctx().raw(), accept_ticket, r), "dsh_reduction_wait(accept)");
        sel_norms[0] = r[0];
        sel_norms[1] = r[1];
      }
    } else {
      update_diff(order_, y_delta_);
      for (size_t j = 0; j < sdiff_.size(); ++j) update_diff_of(sdiff_[j], order_, s_deltas_[j]);  // update_differences_and_integrate_out (bdf.rs:628-643)
      y_.copy_from(y_predict_);
      t_ = t_predict_;
      dy_.copy_from_view(diff_.column(1));
      dy_.mul_assign(scale(1.0 / h_));
    }
    statistics_.number_of_steps += 1;
    jacobian_update_.step();
    prev_error_norm_ = error_norm;
    n_equal_steps_ += 1;

    if (n_equal_steps_ > order_) {  // order selection (bdf.rs:1489-1563)
      const int order = order_;
      const double inf = std::numeric_limits<double>::infinity();
      double error_m_norm = inf, error_p_norm = inf;
      if (order > 1) error_m_norm = std::fmax(0.0, ((fused_ || fused_la_) ? sel_norms[0] : diff_.column(order).squared_norm(y_, pr_.atol, pr_.rtol)) * error_const2_[(size_t)order - 1]);
      if (order < MAX_ORDER) error_p_norm = std::fmax(0.0, ((fused_ || fused_la_) ? sel_norms[1] : diff_.column(order + 2).squared_norm(y_, pr_.atol, pr_.rtol)) * error_const2_[(size_t)order + 1]);
      if (pr_.sens && pr_.sens_error_control) {  // predict_error_control with the augmented system (bdf.rs:908-919)
        for (size_t j = 0; j < sdiff_.size(); ++j) {
          if (order > 1) error_m_norm = std::fmax(error_m_norm, sdiff_[j].column(order).squared_norm(s_[j], pr_.sens_atol, pr_.sens_rtol) * error_const2_[(size_t)order - 1]);
          if (order < MAX_ORDER) error_p_norm = std::fmax(error_p_norm, sdiff_[j].column(order + 2).squared_norm(s_[j], pr_.sens_atol, pr_.sens_rtol) * error_const2_[(size_t)order + 1]);
        }
      }
      const double pi_i = pr_.ode_options.pi_control_integral, pi_p = pr_.ode_options.pi_control_proportional;
      const double factors[3] = {pi_controller_raw(error_m_norm, prev_error_norm_, pi_i, pi_p, order), pi_controller_raw(error_norm, prev_error_norm_, pi_i, pi_p, order + 1),
                                 pi_controller_raw(error_p_norm, prev_error_norm_, pi_i, pi_p, order + 2)};
      int max_index = 0;  // Iterator::max_by keeps the last maximum
      for (int k = 1; k < 3; ++k) if (factors[k] >= factors[max_index]) max_index = k;
      const int new_order = max_index == 0 ? order - 1 : (max_index == 1 ? order : order + 1);
      order_ = new_order;
      if (max_index != 1) { u_ = compute_r(new_order, 1.0); prediction_valid_ = false; prelaunch_valid_ = false; }
      double factor = safety * factors[max_index];
      if (factor > maximum_timestep_growth_) factor = maximum_timestep_growth_;
      if (factor < minimum_timestep_shrink_) factor = minimum_timestep_shrink_;
      if (factor >= minimum_timestep_growth_ || factor <= maximum_timestep_shrink_ || max_index == 0 || max_index == 2) {
        double new_h = update_step_size(factor);
        jacobian_updates(new_h * alpha_[(size_t)new_order], SolverState::StepSuccess);
      }
    }

    if (root_finder_) {
      auto interp = [&](double tt, HipVec& yy) { interpolate_inplace(tt, yy); };
      auto ret = root_finder_->check_root(interp, *pr_.eqn, y_, t_);
      if (ret) { root_time = ret->first; root_index = ret->second; return OdeSolverStopReason::RootFound; }
    }
    if (tstop_) {
      if (handle_tstop(*tstop_)) return OdeSolverStopReason::TstopReached;
    }
    return OdeSolverStopReason::InternalTimestep;
  }

  void set_stop_time(double tstop) override {  // bdf.rs:1591-1600
    tstop_ = tstop;
    if (handle_tstop(tstop)) { tstop_.reset(); throw DSH_ODE_ERR(StopTimeAtCurrentTime); }
  }

  void interpolate_inplace(double t, HipVec& y) const override {  // bdf.rs:1081-1108
    if (y.len() != y_.len()) throw DSH_ODE_ERR(InterpolationVectorWrongSize);
    if (is_state_modified_) {
      if (t == t_) { y.copy_from(y_); return; }
      throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep);
    }
    const bool is_forward = h_ > 0.0;
    if ((is_forward && t > t_) || (!is_forward && t < t_)) throw DSH_ODE_ERR(InterpolationTimeAfterCurrentTime);
    // interpolate_from_diff (bdf.rs:767-782)
    double time_factor = 1.0;
    y.copy_from_view(diff_.column(0));
    for (int i = 0; i < order_; ++i) {
      double i_t = (double)i;
      time_factor *= (t - (t_ - h_ * i_t)) / (h_ * (1.0 + i_t));
      y.axpy_v(time_factor, diff_.column(i + 1), 1.0);
    }
  }
  void interpolate_dy_inplace(double t, HipVec& dy) const {  // bdf.rs:784-811
    double pi = 1.0, d_pi = 0.0;
    dy.fill(0.0);
    for (int i = 0; i < order_; ++i) {
      double i_t = (double)i, denom = h_ * (1.0 + i_t);
      double w = (t - (t_ - h_ * i_t)) / denom, dw = 1.0 / denom;
      double new_d_pi = d_pi * w + pi * dw;
      pi *= w;
      d_pi = new_d_pi;
      dy.axpy_v(d_pi, diff_.column(i + 1), 1.0);
    }
  }
  void apply_reset() override {
    if (pr_.sens) throw LaError(DSH_E_UNSUPPORTED, "apply_reset with forward sensitivities is not supported by the HIP backend");
    HipVec y_out = HipVec::zeros(n(), ctx());
    pr_.eqn->reset_call_inplace(y_, t_, y_out);
    y_.copy_from(y_out);
    is_state_modified_ = true;  // already set by state_mut_back (bdf.rs:1260)
    if (pr_.eqn->has_mass()) {  // state.rs:297-300
      StateCommon sc; sc.y = std::move(y_); sc.dy = std::move(dy_); sc.t = t_; sc.h = h_;
      try { set_consistent(sc, pr_, true); } catch (...) { y_ = std::move(sc.y); dy_ = std::move(sc.dy); throw; }
      y_ = std::move(sc.y); dy_ = std::move(sc.dy);
      return;
    }
    pr_.eqn->rhs_call_inplace(y_, t_, y_out);
    dy_.copy_from(y_out);
  }
  // state_mut_back (bdf.rs:1232-1262): move the state to an interpolated time inside the last step
  void state_mut_back(double t) override {
    if (pr_.sens) throw LaError(DSH_E_UNSUPPORTED, "state_mut_back with forward sensitivities is not supported by the HIP backend");
    if (is_state_modified_) { if (t != t_) throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep); return; }
    const bool is_forward = h_ > 0.0;
    if ((is_forward && t > t_) || (!is_forward && t < t_)) throw DSH_ODE_ERR(InterpolationTimeAfterCurrentTime);
    HipVec ynew = HipVec::zeros(n(), ctx()), dynew = HipVec::zeros(n(), ctx());
    interpolate_inplace(t, ynew);
    interpolate_dy_inplace(t, dynew);
    y_.copy_from(ynew);
    dy_.copy_from(dynew);
    t_ = t;
    is_state_modified_ = true;
  }

  const HipVec& y() const override { return y_; }
  const HipVec& dy() const override { return dy_; }
  double t() const override { return t_; }
  double h() const override { return h_; }
  int order() const override { return order_; }
  const OdeSolverStatistics& get_statistics() const override { return statistics_; }
  const OdeSolverProblem& problem() const override { return pr_; }
  const HipMat& diff() const { return diff_; }
  bool is_fused() const { return fused_; }
  void set_fuse_accept(bool v) { fuse_accept_ = v; }

 private:
  int64_t n() const { return pr_.eqn->nstates(); }
  int64_t nb() const { return pr_.context().nbatch(); }
  const HipContext& ctx() const { return pr_.context(); }

  void initialise_diff_to_first_order() {  // bdf_state.rs:72-78
    order_ = 1;
    diff_.column_mut(0).copy_from(y_);
    diff_.column_mut(1).copy_from(dy_);
    diff_.column_mut(1).mul_assign(scale(h_));
  }

  // NewtonNonlinearSolver::reset_jacobian(op, state.y, state.t): assemble M - cJ and factor
  void reset_jacobian() {
    first_linearisation_pending_ = false;  // any factorisation supersedes the constructor's (its Jacobian is still marked stale, so it is evaluated here)
    prelaunch_valid_ = false;  // a pre-launched Newton iteration used the old factors
    if (fused_) {
      const bool recompute = op_.jacobian_is_stale();
      if (recompute) { pr_.eqn->rhs_statistics.number_of_matrix_evals++; pr_.eqn->rhs_statistics.number_of_jac_muls += n(); }
      DSH_PROF(jac, check(dsh_jac_factor(ctx().raw(), model_, model_size_, nb(), t_, op_.c(), y_.ptr(), pr_.eqn->params().ptr(), recompute ? 1 : 0, op_.rhs_jac().ptr(),
                           op_.mass_jac().ptr(), nonlinear_solver_.linear_solver().raw()), "dsh_jac_factor"));
      op_.clear_jacobian_is_stale();
      nonlinear_solver_.mark_jacobian_set();
    } else {
      nonlinear_solver_.reset_jacobian(op_, y_, t_);
    }
  }

  void jacobian_updates(double c, SolverState state) {  // bdf.rs:465-506
    bool did_update = false;
    if (jacobian_update_.check_rhs_jacobian_update(c, state)) {
      op_.set_jacobian_is_stale();
      reset_jacobian();
      jacobian_update_.update_rhs_jacobian(c);
      jacobian_update_.update_jacobian(c);
      convergence_.reset_eta();
      did_update = true;
    } else if (jacobian_update_.check_jacobian_update(c, state)) {
      reset_jacobian();
      jacobian_update_.update_jacobian(c);
      convergence_.reset_eta();
      did_update = true;
    }
    if (did_update) record_linear_solver_setup(statistics_, state);
  }

  double update_step_size(double factor, bool ignore_too_small = false) {  // bdf.rs:508-577
    const double new_h = factor * h_;
    n_equal_steps_ = 0;
    prediction_valid_ = false;
    prelaunch_valid_ = false;
    const int order = order_;
    std::vector<double> r = compute_r(order, factor);
    std::vector<double> ru = mat_mul_small(r, u_, order + 1);
    // D[:,0..=order] <- D[:,0..=order] * RU into diff_tmp, then swap(diff, diff_tmp) — including the reference's stale-column quirk
    if (fused_ || fused_la_) {
      check(dsh_bdf_prepare_step(ctx().raw(), n(), nb(), order, diff_.ptr(), diff_tmp_.ptr(), ru.data(), gamma_.data(), alpha_[(size_t)order], nullptr, nullptr),
            "dsh_bdf_prepare_step(rescale)");
    } else {
      HipMat ru_dev = HipMat::from_vec(order + 1, order + 1, ru, ctx().clone_with_nbatch(1));
      diff_tmp_.columns_mut(0, order + 1).gemm_vo(1.0, diff_.columns(0, order + 1), ru_dev, 0.0);
    }
    diff_.swap(diff_tmp_);
    if (pr_.sens) {  // bdf.rs:546-553: every sdiff through the SAME scratch matrix, then set_c on the sensitivity operator
      HipMat ru_dev = HipMat::from_vec(order + 1, order + 1, ru, ctx().clone_with_nbatch(1));
      for (HipMat& sd : sdiff_) {
        diff_tmp_.columns_mut(0, order + 1).gemm_vo(1.0, sd.columns(0, order + 1), ru_dev, 0.0);
        sd.swap(diff_tmp_);
      }
      s_c_ = new_h * alpha_[(size_t)order];
    }
    op_.set_c(new_h, alpha_[(size_t)order]);
    h_ = new_h;
    convergence_.reset_eta_timestep_change();
    if (!ignore_too_small && std::fabs(h_) < minimum_timestep_) throw DSH_ODE_ERR(StepSizeTooSmall);
    return new_h;
  }

  void update_diff_of(HipMat& diff, int order, const HipVec& d) {  // _update_diff on a sensitivity difference array
    d_tmp_.copy_from(d);
    d_tmp_.sub_assign(diff.column(order + 1));
    diff.column_mut(order + 2).copy_from(d_tmp_);
    diff.column_mut(order + 1).copy_from(d);
    for (int i = order; i >= 0; --i) diff.column_axpy(1.0, i + 1, i);
  }
  void update_diff(int order, const HipVec& d) {  // bdf.rs:646-664 (trait mode)
    d_tmp_.copy_from(d);
    d_tmp_.sub_assign(diff_.column(order + 1));
    diff_.column_mut(order + 2).copy_from(d_tmp_);
    diff_.column_mut(order + 1).copy_from(d);
    for (int i = order; i >= 0; --i) diff_.column_axpy(1.0, i + 1, i);
  }

  void predict_forward() {  // bdf.rs:674-692
    if ((fused_ || fused_la_) && prediction_valid_) {
      prediction_valid_ = false;  // produced by the accept launch of the previous step (same D, order and h => same bits)
    } else if (fused_ || fused_la_) {
      DSH_PROF(prepare, check(dsh_bdf_prepare_step(ctx().raw(), n(), nb(), order_, diff_.ptr(), diff_tmp_.ptr(), nullptr, gamma_.data(), alpha_[(size_t)order_], y_predict_.ptr(),
                                 op_.psi_neg_y0().ptr()), "dsh_bdf_prepare_step"));
    } else {
      y_predict_.fill(0.0);
      for (int i = 0; i <= order_; ++i) y_predict_.add_assign(diff_.column(i));
      op_.set_psi_and_y0(diff_, gamma_, alpha_, order_, y_predict_);
    }
    t_predict_ = t_ + h_;
  }

  // newton_iteration + NoLineSearch::take_optimal_step with the whole iteration in one launch (dsh_bdf_newton_iter)
  // newton_iteration + NoLineSearch::take_optimal_step with each iteration in ONE launch (dsh_bdf_newton_iter), pipelined: while the host
  // evaluates the convergence test of iteration k, iteration k+1 is already running speculatively into the other iterate buffer
  // (it k reads buf[(k-1)%2] — the predictor for k = 1 — and writes buf[k%2]).  If iteration k turns out to have converged / diverged
  // the speculative launch is simply ignored; otherwise its result is the next one needed.  Every iterate is the same function of the
  // same inputs as in the sequential algorithm, so results stay bit-identical; only the GPU idle time of the host round trip goes away.
  // Enqueue one launch of `nit` consecutive Newton iterations for the step ending at t_pred, starting from `y_in`; the iterates go to
  // group buffer `g` (nit_ * n * nb doubles each).
  void launch_newton(int nit, const double* y_in, int g, double t_pred, int64_t* ticket) {
    DSH_PROF(launch_newton, check(dsh_bdf_newton_iter_async(ctx().raw(), model_, model_size_, nb(), t_pred, op_.c(), nit, y_in, ybuf_[g].ptr(), op_.psi_neg_y0().ptr(),
                                    pr_.eqn->params().ptr(), nonlinear_solver_.linear_solver().raw(), y_predict_.ptr(), y_.ptr(), pr_.atol.ptr(),
                                    pr_.atol.nb(), pr_.rtol, ticket), "dsh_bdf_newton_iter_async"));
  }

  // newton_iteration + NoLineSearch::take_optimal_step (diffsol-nl/src/newton.rs:13-36, line_search.rs:48-69) with nit_ iterations per
  // launch.  The host walks the returned norms through the reference's Convergence state machine in order and stops at the first
  // Converged / Diverged verdict; iterates computed beyond that point are discarded speculative work.  Every iterate is the same function
  // of the same inputs as in the one-iteration-at-a-time algorithm, so results and counters stay bit-identical.
  NlError newton_fused(double& err_sq_out) {
    if (!nonlinear_solver_.is_jacobian_set()) return NlError::JacobianNotReset;
    convergence_.reset();
    const int max_iter = convergence_.max_iter();
    const int64_t stride = n() * nb();
    int done = 0, g = 0;
    const double* src = y_predict_.ptr();
    bool use_prelaunch = prelaunch_valid_;
    prelaunch_valid_ = false;
    while (done < max_iter) {
      const int nit = std::min(nit_, max_iter - done);
      int64_t ticket = 0;
      double out[12];
      int rc;
      if (use_prelaunch && nit == prelaunch_nit_) {  // enqueued right after the previous step's accept launch (group buffer 0)
        ticket = prelaunch_ticket_;
        DSH_PROF(wait_newton, rc = dsh_reduction_wait(ctx().raw(), ticket, out));
        if (rc == DSH_E_STALE) {  // its records were recycled by other reductions (e.g. root finding): redo it — same inputs, same bits
          launch_newton(nit, src, g, t_predict_, &ticket);
          rc = dsh_reduction_wait(ctx().raw(), ticket, out);
        }
      } else {
        launch_newton(nit, src, g, t_predict_, &ticket);
        DSH_PROF(wait_newton, rc = dsh_reduction_wait(ctx().raw(), ticket, out));
      }
      use_prelaunch = false;
      check(rc, "dsh_reduction_wait");
      for (int i = 0; i < nit; ++i) {
        pr_.eqn->rhs_statistics.number_of_calls++;  // iteration i evaluated f(y) once for every system
        if (out[3 * i + 2] != 0.0) return NlError::LuSolveFailed;
        ConvergenceStatus st = convergence_.check_new_iteration(std::sqrt(out[3 * i + 0]));
        err_sq_out = out[3 * i + 1];
        y_new_ = ybuf_[g].ptr() + (int64_t)i * stride;
        if (st == ConvergenceStatus::Converged) return NlError::Ok;
        if (st == ConvergenceStatus::Diverged) return NlError::NewtonDiverged;
      }
      done += nit;
      src = ybuf_[g].ptr() + (int64_t)(nit - 1) * stride;
      g ^= 1;
    }
    return NlError::NewtonMaxIterations;
  }

  // returns true when tstop has been reached (bdf.rs:694-731)
  bool handle_tstop(double tstop) {
    const double eps = std::numeric_limits<double>::epsilon();
    const double troundoff = 100.0 * eps * (std::fabs(t_) + std::fabs(h_));
    if (std::fabs(t_ - tstop) <= troundoff) { tstop_.reset(); return true; }
    if ((h_ > 0.0 && tstop < t_ - troundoff) || (h_ < 0.0 && tstop > t_ + troundoff)) { tstop_.reset(); throw DSH_ODE_ERR(StopTimeBeforeCurrentTime); }
    if ((h_ > 0.0 && t_ + h_ > tstop + troundoff) || (h_ < 0.0 && t_ + h_ < tstop - troundoff)) {
      const double factor = (tstop - t_) / h_;
      (void)update_step_size(factor, /*ignore_too_small=*/true);
    }
    return false;
  }

  const OdeSolverProblem& pr_;
  NewtonNonlinearSolver nonlinear_solver_;
  NoLineSearch line_search_;
  Convergence convergence_;
  BdfCallable op_;
  int n_equal_steps_ = 0;
  HipVec y_delta_, y_predict_, d_tmp_;
  HipVec ybuf_[2];                 // fused mode: two groups of nit_ Newton iterates each (ping-pong between launches)
  int nit_ = 3;                    // Newton iterations per launch (DSH_NEWTON_NIT = 1..4)
  int prelaunch_nit_ = 0;
  const double* y_new_ = nullptr;  // fused mode: buffer holding the converged iterate
  HostProfile prof_;
  bool pipeline_ = true;           // speculative Newton pipelining (DSH_NEWTON_PIPELINE=0 disables)
  bool fuse_accept_ = false;       // accept + prelaunched Newton iterations in one launch (DSH_FUSE_ACCEPT=1 enables; pays only when GPU-bound)
  bool prelaunch_valid_ = false;   // iteration 1 of the next step is already in flight (ticket below)
  int64_t prelaunch_ticket_ = 0;
  double t_predict_ = 0.0;
  HipMat diff_, diff_tmp_;
  bool first_linearisation_pending_ = false;
  double first_linearisation_c_ = 0.0;
  std::vector<double> u_, alpha_, gamma_, error_const2_;
  OdeSolverStatistics statistics_;
  // BdfState (bdf_state.rs:13-37)
  int order_ = 1;
  HipVec y_, dy_;
  double t_ = 0.0, h_ = 0.0;
  std::optional<double> tstop_;
  std::optional<RootFinder> root_finder_;
  bool is_state_modified_ = false;
  JacobianUpdate jacobian_update_;
  double minimum_timestep_, maximum_timestep_growth_, minimum_timestep_growth_, maximum_timestep_shrink_, minimum_timestep_shrink_;
  int maximum_error_test_failures_, maximum_newton_fails_;
  std::optional<double> prev_error_norm_;
  // forward sensitivities
  std::vector<HipVec> s_, ds_, s_deltas_;
  std::vector<HipMat> sdiff_;
  HipVec s_predict_, s_psi_neg_y0_, s_tmp_, sens_y_;
  HipMat sens_mat_;
  double s_c_ = 0.0;
  bool fused_ = false;
  bool fused_la_ = false;  // model-independent difference-array kernels for run-time-sized models (large ensembles)
  bool prediction_valid_ = false;
  int model_ = -1;
  int64_t model_size_ = 0;
};

}  // namespace diffsol_hip
