// One WORKGROUP per ensemble member: variable-order BDF for run-time-sized DENSE models with 64 < n <= 320 (launch code: dsh_wave_member.hip).
// Closes the gap between the wavefront-per-member kernel (n <= 64: a matrix row per lane, in registers) and the host-driven lock-step path (VERDICT r3 missing 1:
// the reference's Bdf::step is size-generic, crates/diffsol/src/ode_solver/bdf.rs:1277-1589, and its own published benchmark sizes are n = 30 / 300).
//   * thread t of the 128 / 192 threads holds component t of the state, of the prediction, of psi and its row of the difference array — as a lane does in
//     k_bdf_wave_member; every scalar of Bdf::step is computed redundantly by all threads from the same norms, so the control flow is workgroup-uniform;
//   * M - c J and its LU factors live in the 160 KB LDS (column-major, odd pitch: column reads by consecutive rows and row reads by consecutive columns are both
//     conflict-free), rows PHYSICALLY at their final positions (an interchange is one LDS swap per column, done by n threads at once), so the triangular solves
//     need no position look-ups: wavefront b owns positions 64 b .. 64 b + 63, eliminates them among its own lanes with v_readlane (no barrier), publishes the 64
//     finished unknowns through LDS and the other wavefronts apply them in index order — two barriers per 64 unknowns instead of one per unknown;
//   * the cached Jacobian of a member sits in global scratch (n^2 doubles per member, read once per refactorisation, coalesced; L2 / MALL resident);
//   * norms: per-row terms through LDS, summed by every thread in index order (the oracle's sequential sum).
// Per element the arithmetic and its order are those of wave_lu_factor_rows / wave_lu_solve_rows (dsh_lu_wave.hpp) and of the oracle: l = a (1 / pivot),
// a_rc = (-u_kc) l_rk + a_rc, first largest magnitude wins the pivot search, column-oriented substitutions — bit-identical results (tests/test_gpu_team_member.py).
// The BDF logic below is k_bdf_wave_member's text with the wavefront primitives replaced (generated from it once, then maintained here).
#pragma once
#include "dsh_wave_member_kernel.hpp"
#include "dsh_team_reg_lu.hpp"

namespace dsh {

constexpr int kTeamMaxN = 320;     // one workgroup per member up to here (round 5: 140 < n <= 320 with the factors in global scratch)
constexpr int kTeamLdsMaxN = 140;  // up to here the factors fit the 160 KB of LDS
// wavefronts of the workgroup (a thread per row): 2 (n <= 128), 3 (n <= 140) with the factors in LDS; 4 (n <= 256), 5 (n <= 320) with the factors in the member's global
// scratch behind its cached Jacobian — the same code on another address space (L2 / Infinity-Cache resident: 256 members in flight x 0.8 MB), for the models the
// reference's own benchmark family reaches (robertson_ode x 100: n = 300, book/src/benchmarks/python_results.csv:12-13)
__host__ __device__ constexpr int team_waves(int n) { return n <= 128 ? 2 : (n <= kTeamLdsMaxN ? 3 : (n <= 256 ? 4 : 5)); }
__host__ __device__ constexpr bool team_global_factors(int waves) { return waves >= 4; }
// pitch of the factors: ONE odd value per workgroup shape, a compile-time constant — with a run-time pitch every
// LDS access of the factorisation carries its own index arithmetic (scripts/team_member_prof.sh: 620 cycles per eight columns of a pivot step's update)
__host__ __device__ constexpr int team_pitch_w(int waves) { return waves <= 2 ? 129 : (waves == 3 ? 141 : 64 * waves + 1); }
__host__ __device__ inline size_t team_lds_doubles(int n, int waves) {
  return (size_t)(3 * 64 * waves + 2 * waves + 32 * waves) + (team_global_factors(waves) ? (size_t)0 : (size_t)n * team_pitch_w(waves));
}
// the register-resident LU (64 < n <= 128, dsh_team_reg_lu.hpp): xs | xs2 | ps | cand of the integrator, then the LU's workspace
__host__ __device__ constexpr size_t team_rl_lds_doubles(int NL) { return (size_t)(3 * 128 + 4) + (size_t)trg_lds_doubles(NL); }
// global scratch per member: the cached Jacobian (n^2) and, beyond the LDS sizes, the factors (n x pitch)
__host__ __device__ inline size_t team_scratch_doubles(int n, int waves) { return (size_t)n * n + (team_global_factors(waves) ? (size_t)n * team_pitch_w(waves) : (size_t)0); }

// sum of n terms held in LDS, in index order, the reads of eight terms issued together (one read per round trip: 94 cycles per term)
// (the next eight are requested before the current eight are added: the additions are one dependent chain, the reads need not wait inside it — a norm of n = 120 terms
// the reads of a batch no longer stand between two additions; there are five to eight norms per step)
__device__ __forceinline__ double team_seq_sum(const double* __restrict__ red, int n) {
  double acc = 0.0;
  int i = 0;
  if (n >= 8) {
    double cur[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cur[u] = red[u];
    for (; i + 16 <= n; i += 8) {
      double nxt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) nxt[u] = red[i + 8 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += cur[u];
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += cur[u];
    i += 8;
  }
  for (; i < n; ++i) acc += red[i];
  return acc;
}

#ifdef DSH_TEAM_MEMBER_PROF
__device__ unsigned long long g_tlu[4];  // thread 0 of workgroup 0: cycles in the pivot search, the interchange, the update (its barrier = the slowest thread)
#define TLU_MARK(q) if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_tlu[q] += now_ - tlu_t0; tlu_t0 = now_; }
#else
#define TLU_MARK(q)
#endif
// Pivot of elimination step k over the W wavefronts (largest magnitude, smallest row on ties): the wavefront's winning lane publishes the SIGNED entry, so every
// thread has the pivot value from LDS and nobody reads A[k * P + p] while thread k interchanges that very entry (a wavefront that came late read the
// already swapped value — with the factors in global scratch the window was two memory round trips wide).  Returns p (n: no candidate — a NaN column).
template <int W>
__device__ __forceinline__ int team_pivot(const double* __restrict__ A, int P, int n, int k, int ln, bool rowlive, double* __restrict__ cand, double& diag) {
  const int wave = ln >> 6, lane = ln & 63;
  double best = -1.0, vs = 0.0;
  int p = n;
  if (rowlive && ln >= k) { vs = A[k * P + ln]; const double v = fabs(vs); if (v > best) { best = v; p = ln; } }
  group_argmax(best, p, 64);  // largest magnitude, smallest row on ties, over this wavefront
  if (p < n ? ln == p : lane == 0) { cand[2 * wave] = vs; cand[2 * wave + 1] = (double)p; }
  __syncthreads();
  double s0 = cand[0];
  int p0 = (int)cand[1];
  double b0 = p0 < n ? fabs(s0) : -1.0;
#pragma unroll
  for (int w = 1; w < W; ++w) {
    const double sw = cand[2 * w];
    const int pw = (int)cand[2 * w + 1];
    const double bw = pw < n ? fabs(sw) : -1.0;
    if (bw > b0 || (bw == b0 && pw < p0)) { b0 = bw; p0 = pw; s0 = sw; }
  }
  diag = p0 < n ? s0 : A[k * P + k];  // NaN column: keep the diagonal like the sequential scan (no interchange follows, nothing to race with)
  return p0 < n ? p0 : k;
}
// LU of the n x n matrix in LDS (A[c * P + r], thread t = row t) with partial pivoting and physical row interchanges; perm[k] = original row at position k.
// Workgroup-uniform control flow; all W wavefronts must call it together.
template <int W>
__device__ __forceinline__ void team_lu_factor_panel(double* __restrict__ A, int P, int n, int ln, bool rowlive, double* __restrict__ cand, int* __restrict__ perm, bool& singular);
template <int W>
__device__ __forceinline__ void team_lu_factor(double* __restrict__ A, int P, int n, int ln, bool rowlive, double* __restrict__ cand, int* __restrict__ perm, bool& singular) {
  if constexpr (team_global_factors(W)) { team_lu_factor_panel<W>(A, P, n, ln, rowlive, cand, perm, singular); return; }  // factors in global scratch: one round trip of the trailing matrix per panel
  perm[ln] = ln;
  singular = false;
  __syncthreads();  // every row of A is in LDS, perm is the identity
#ifdef DSH_TEAM_MEMBER_PROF
  unsigned long long tlu_t0 = clock64();
#endif
  for (int k = 0; k < n; ++k) {
    double diag;
    int p = team_pivot<W>(A, P, n, k, ln, rowlive, cand, diag);
    TLU_MARK(0)
    const bool elim = diag != 0.0;  // a zero pivot leaves the rows where they are (lu_factor_reg does the same)
    if (!elim) { singular = true; p = k; }
    if (elim && p != k) {  // interchange rows k and p: thread c takes column c
      if (ln < n) { const double u = A[ln * P + k]; A[ln * P + k] = A[ln * P + p]; A[ln * P + p] = u; }
      if (ln == 0) { const int q = perm[k]; perm[k] = perm[p]; perm[p] = q; }
    }
    __syncthreads();
    TLU_MARK(1)
    if (elim && rowlive && ln > k) {
      const double l = A[k * P + ln] * (1.0 / diag);
      A[k * P + ln] = l;
      // eight columns at a time, their loads issued together: one column per LDS round trip costs 10 000 cycles per pivot step at n = 120 (scripts/team_member_prof.sh)
      int c = k + 1;
      for (; c + 8 <= n; c += 8) {
        double pk[8], mine[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { pk[u] = A[(c + u) * P + k]; mine[u] = A[(c + u) * P + ln]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) A[(c + u) * P + ln] = (-pk[u]) * l + mine[u];
      }
      for (; c < n; ++c) A[c * P + ln] = (-A[c * P + k]) * l + A[c * P + ln];
    }
    __syncthreads();
    TLU_MARK(2)
  }
}

// The same factorisation in PANELS of kTeamPanel pivots (LAPACK getrf's blocking), for the factors in GLOBAL scratch (n > 140): the unblocked form streams the whole
// trailing matrix through the memory system once per pivot — n^3 / 3 x 24 bytes = 216 MB per factorisation at n = 300, nothing of it cached with thousands of members
// in flight: the route was HBM-bound (profiles/r05_member_lanes.md).  Per element the operations of the unblocked elimination in their order — a_rc takes
// (-u_kc) l_rk + a_rc for k = 0, 1, 2, ... whatever the grouping: the same bits.
//   1. the panel's pivots are searched, interchanged (whole rows, as before) and eliminated inside the panel's own columns only;
//   2. U12: the pivot rows' entries in every later column take the panel's earlier pivots — thread j does column k1 + j by itself and writes the final entries back;
//   3. the rows below the panel update every later column with all of the panel's pivots at once: the trailing matrix makes one round trip per PANEL.
// (In LDS this form measured no gain — 0.349 against 0.351 s at n = 120 — and is not used there.)
constexpr int kTeamPanel = 8;  // (16 measured 3.56 against 3.61 s at n = 300 and slower at n = 150)
template <int W>
__device__ __forceinline__ void team_lu_factor_panel(double* __restrict__ A, int P, int n, int ln, bool rowlive, double* __restrict__ cand, int* __restrict__ perm, bool& singular) {
  constexpr int PW = kTeamPanel;
  perm[ln] = ln;
  singular = false;
  __syncthreads();  // every row of A is stored, perm is the identity
  for (int k0 = 0; k0 < n; k0 += PW) {
    const int k1 = k0 + PW < n ? k0 + PW : n;
    unsigned elim_mask = 0u;  // pivots of this panel that eliminate (a zero pivot leaves the rows where they are and eliminates nothing, lu_factor_reg does the same)
    for (int k = k0; k < k1; ++k) {
      double diag;
      int p = team_pivot<W>(A, P, n, k, ln, rowlive, cand, diag);
      const bool elim = diag != 0.0;
      if (!elim) { singular = true; p = k; }
      if (elim) elim_mask |= 1u << (k - k0);
      if (elim && p != k) {  // interchange rows k and p: thread c takes column c
        if (ln < n) { const double u = A[ln * P + k]; A[ln * P + k] = A[ln * P + p]; A[ln * P + p] = u; }
        if (ln == 0) { const int q = perm[k]; perm[k] = perm[p]; perm[p] = q; }
      }
      __syncthreads();
      if (elim && rowlive && ln > k) {
        const double l = A[k * P + ln] * (1.0 / diag);
        A[k * P + ln] = l;
        for (int c = k + 1; c < k1; ++c) A[c * P + ln] = (-A[c * P + k]) * l + A[c * P + ln];
      }
      __syncthreads();
    }
    if (k1 >= n) break;
    // ---- U12: thread j finishes the pivot rows' entries of column k1 + j (row k0 + u takes the pivots k0 + v, v < u, in order); n - k1 <= threads
    {
      const int c = k1 + ln;
      if (c < n) {
        double t[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) t[u] = k0 + u < k1 ? A[c * P + k0 + u] : 0.0;
#pragma unroll
        for (int u = 1; u < PW; ++u) {
          if (k0 + u < k1) {
#pragma unroll
            for (int v = 0; v < u; ++v)
              if (elim_mask & (1u << v)) t[u] = (-t[v]) * A[(k0 + v) * P + k0 + u] + t[u];  // l_{k0+u, k0+v}: the same address in every lane
          }
        }
#pragma unroll
        for (int u = 1; u < PW; ++u) if (k0 + u < k1) A[c * P + k0 + u] = t[u];
      }
    }
    __syncthreads();
    // ---- rows below the panel: every later column takes the panel's pivots in order
    if (rowlive && ln >= k1) {
      double lr[PW];
#pragma unroll
      for (int u = 0; u < PW; ++u) lr[u] = k0 + u < k1 ? A[(k0 + u) * P + ln] : 0.0;
      int c = k1;
      for (; c + 4 <= n; c += 4) {  // four columns at a time, their reads issued together
        double pu[4][PW], mine[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          mine[q] = A[(c + q) * P + ln];
#pragma unroll
          for (int u = 0; u < PW; ++u) pu[q][u] = A[(c + q) * P + (k0 + u < k1 ? k0 + u : k0)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int u = 0; u < PW; ++u)
            if (k0 + u < k1 && (elim_mask & (1u << u))) mine[q] = (-pu[q][u]) * lr[u] + mine[q];
          A[(c + q) * P + ln] = mine[q];
        }
      }
      for (; c < n; ++c) {
        double m1 = A[c * P + ln];
#pragma unroll
        for (int u = 0; u < PW; ++u)
          if (k0 + u < k1 && (elim_mask & (1u << u))) m1 = (-A[c * P + k0 + u]) * lr[u] + m1;
        A[c * P + ln] = m1;
      }
    }
    __syncthreads();
  }
}

// Solve with the factors above: on entry thread t holds component t of the right-hand side, on return unknown t.  xch: T doubles of LDS.  Returns false when a
// pivot is zero (the factorisation recorded it), like wave_lu_solve_rows.  Per element: the column-oriented substitutions of lu_solve_reg / the oracle.
template <int W>
__device__ __forceinline__ bool team_lu_solve(const double* __restrict__ A, int P, int n, int ln, bool rowlive, const int* __restrict__ perm, double* __restrict__ xch,
                                              bool singular, double& v) {
  const int wave = ln >> 6;
  constexpr int SC = 8;  // steps whose factors are read ahead of the dependent chain (32 for the factors in global scratch measured no gain: 3.74 against 3.61 s at n = 300)
  __syncthreads();
  xch[ln] = v;
  __syncthreads();
  v = rowlive ? xch[perm[ln]] : 0.0;  // (P b) at my position
  // L y = P b (unit lower triangle)
#pragma unroll
  for (int B = 0; B < W; ++B) {
    const int k0 = 64 * B;
    if (k0 < n) {
      if (wave == B) {
        for (int kk0 = 0; kk0 < 64; kk0 += SC) {  // the factors of SC steps are read ahead of the chain (they do not depend on it)
          double a8[SC];
#pragma unroll
          for (int u = 0; u < SC; ++u) { const int k = k0 + kk0 + u; a8[u] = (k + 1 < n && rowlive && ln > k) ? A[k * P + ln] : 0.0; }
#pragma unroll
          for (int u = 0; u < SC; ++u) {
            const int k = k0 + kk0 + u;
            if (k + 1 < n) {
              const double coeff = group_bcast<64>(v, kk0 + u);
              if (rowlive && ln > k) v = (-coeff) * a8[u] + v;
            }
          }
        }
      }
      if (B + 1 < W && k0 + 64 < n) {  // rows of later wavefronts: the 64 finished unknowns through LDS, applied in index order
        __syncthreads();
        if (wave == B) xch[ln] = v;
        __syncthreads();
        if (wave > B && rowlive)
          for (int kb = k0; kb < k0 + 64; kb += SC) {
            double x8[SC], a8[SC];
#pragma unroll
            for (int u = 0; u < SC; ++u) { x8[u] = xch[kb + u]; a8[u] = A[(kb + u) * P + ln]; }
#pragma unroll
            for (int u = 0; u < SC; ++u) v = (-x8[u]) * a8[u] + v;
          }
      }
    }
  }
  // U x = y
#pragma unroll
  for (int B = W - 1; B >= 0; --B) {
    const int k0 = 64 * B;
    if (k0 < n) {
      const int k1 = n < k0 + 64 ? n : k0 + 64;
      if (wave == B) {
        for (int kt = k1 - 1; kt >= k0; kt -= SC) {  // SC steps' diagonal and column entries read ahead of the chain
          double d8[SC], a8[SC];
#pragma unroll
          for (int u = 0; u < SC; ++u) { const int k = kt - u; d8[u] = k >= k0 ? A[k * P + k] : 1.0; a8[u] = (k >= k0 && rowlive && ln < k) ? A[k * P + ln] : 0.0; }
#pragma unroll
          for (int u = 0; u < SC; ++u) {
            const int k = kt - u;
            if (k >= k0) {
              const double coeff = group_bcast<64>(v, k - k0) / d8[u];
              if (ln == k) v = coeff;
              else if (rowlive && ln < k) v = (-coeff) * a8[u] + v;
            }
          }
        }
      }
      if (B > 0) {
        __syncthreads();
        if (wave == B) xch[ln] = v;
        __syncthreads();
        if (wave < B) {
          int k = k1 - 1;
          for (; k - (SC - 1) >= k0; k -= SC) {
            double x8[SC], a8[SC];
#pragma unroll
            for (int u = 0; u < SC; ++u) { x8[u] = xch[k - u]; a8[u] = A[(k - u) * P + ln]; }
#pragma unroll
            for (int u = 0; u < SC; ++u) v = (-x8[u]) * a8[u] + v;
          }
          for (; k >= k0; --k) v = (-xch[k]) * A[k * P + ln] + v;
        }
      }
    }
  }
  return !singular;
}

// SENS: forward sensitivities of every parameter alongside, as in k_bdf_wave_member<.., SENS> (run-time-compiled ODE models without root functions; bdf.rs:370-432, :934-989)
// -DDSH_TEAM_MEMBER_PROF: thread 0 of workgroup 0 accumulates the cycles of the phases of its member's solve and prints them (scripts/team_member_prof.sh)
#ifdef DSH_TEAM_MEMBER_PROF
#define TMP_DECL unsigned long long tmp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmp_t0 = clock64(), tmp_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long tmp_begin = tmp_t0;
#define TMP_MARK(k) { const unsigned long long now_ = clock64(); tmp_acc[k] += now_ - tmp_t0; tmp_cnt[k] += 1; tmp_t0 = now_; }
#define TMP_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) printf("team member prof (cycles, member 0, n = %d): total %llu | jacobian %llu (%llu) | factor %llu (%llu) | rhs %llu (%llu) | solve %llu (%llu) | norm %llu (%llu) | step-size change %llu (%llu) | rest %llu\n", n, clock64() - tmp_begin, tmp_acc[0], tmp_cnt[0], tmp_acc[1], tmp_cnt[1], tmp_acc[2], tmp_cnt[2], tmp_acc[3], tmp_cnt[3], tmp_acc[4], tmp_cnt[4], tmp_acc[5], tmp_cnt[5], tmp_acc[7]); if (blockIdx.x == 0 && threadIdx.x == 0) printf("  inside the factorisations (cycles): pivot search %llu | interchange %llu | update %llu\n", g_tlu[0], g_tlu[1], g_tlu[2]);
#else
#define TMP_DECL
#define TMP_MARK(k)
#define TMP_PRINT
#endif
// RL > 0: 64 < n <= RL <= 128 with the factors in REGISTERS (dsh_team_reg_lu.hpp; W = 2): 256 threads, thread t and thread t + 128 both carry row t & 127 of the
// integrator's vectors (the same values, computed twice) and one column half each of the Jacobian and of M - c J
template <int W, bool SENS, int RL>
__device__ __forceinline__ void bdf_team_member_body(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, int atol_broadcast,
                                                       const WaveMemberConsts* __restrict__ Cp, const double* __restrict__ t_eval, double* __restrict__ jac_scratch, double* __restrict__ y_out,
                                                       int32_t* __restrict__ stats_out, int32_t* __restrict__ status_out, double* __restrict__ t_root_out,
                                                       int32_t* __restrict__ root_idx_out, int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  static_assert(!kWmHasMass, "the workgroup-per-member kernel takes identity-mass models");
  constexpr int T = 64 * W;            // threads = rows (n <= T)
  extern __shared__ double lds[];      // xs[T] | xs2[T] | ps[T] | cand[2 * W] | perm[T] (int) | A[n][P]  (team_lds_doubles)
  double* xs = lds;                    // the published state vector (model evaluation)
  double* xs2 = lds + T;               // the right-hand side / substitution exchange of the LU solve; between solves the per-row terms of a norm
  double* red = xs2;
  double* ps = lds + 2 * T;
  double* cand = lds + 3 * T;          // pivot candidates of the wavefronts: value, row  (ps: up to T parameters — gaussian_decay has one per state)
  int* perm = reinterpret_cast<int*>(lds + 3 * T + 2 * W);  // original row at every position of P A = L U
  constexpr int P = team_pitch_w(W);
  double* sJ = jac_scratch + (size_t)blockIdx.x * team_scratch_doubles(Cp->n, W);  // the member's cached Jacobian: global scratch (L2 / MALL resident), entry (ln, j) at j * n + ln
  // the LU factors of M - c J, column-major, pitch P, rows at their final positions: in LDS, or (n > 140) in the member's global scratch behind the Jacobian
  double* A = team_global_factors(W) ? sJ + (size_t)Cp->n * Cp->n : lds + 3 * T + 2 * W + T / 2;
  double* wk = lds + 3 * T + 2 * W;  // RL: the workspace of the register-resident LU
  double a_rl[RL > 0 ? 64 : 1], rl_d = 1.0, rl_r = 1.0;  // RL: my half of my row of M - c J / of its factors; the diagonal of my position and div_refined_rcp of it
  const WaveMemberConsts& C = *Cp;
  const dsh_adaptive_options& o = C.r.o;
  const bool det = o.deterministic_pow != 0;
  const int n = C.n, model = C.model;
  const int64_t b = blockIdx.x;
  constexpr int RBN = RL > 0 ? trg_rbn(RL) : 2;  // RL <= 64: one row block (128 threads: thread t and t + 64 carry row t & 63)
  const int ln = RL > 0 ? (int)(threadIdx.x & (64 * RBN - 1)) : (int)threadIdx.x;
  const int half = RL > 0 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> (5 + RBN))) : 0;
  const bool lead = threadIdx.x == 0;
  const bool rowlive = ln < n;
  const double rtol = C.r.rtol;
  const double atol = rowlive ? (atol_broadcast ? atol_g[ln] : atol_g[(int64_t)ln * nb + b]) : 1.0;
  if (ln < C.np) ps[ln] = p_g[(int64_t)ln * nb + b];
  __syncthreads();
  auto Pf = [&](int64_t k) { return ps[k]; };
  auto Xf = [&](int64_t k) { return xs[k]; };
  auto V0 = [&](int64_t) { return 0.0; };
  // component `ln` of f(x, t); x is published through LDS
  auto rhs_of = [&](double x_mine, double tt) __attribute__((always_inline)) -> double {
    __syncthreads();
    xs[ln] = x_mine;
    __syncthreads();
    return rowlive ? wm_component(model, (int64_t)n, tt, (int64_t)ln, Xf, V0, Pf, false) : 0.0;
  };
  // weighted mean square of a distributed vector: (1/n) sum_i (v_i / (|w_i| rtol + atol_i))^2, summed in index order
  auto wms_wave = [&](double v_mine, double w_mine) __attribute__((always_inline)) -> double {
    const double term = rowlive ? v_mine / (fabs(w_mine) * rtol + atol) : 0.0;
    __syncthreads();
    red[ln] = term * term;
    __syncthreads();
    return team_seq_sum(red, n) / (double)n;  // every thread the same sequential sum (Vector::squared_norm's order)
  };

  // ------------------------------------------------------------ new_and_consistent (identity mass: nothing to make consistent) + set_step_size
  int32_t status = kRsOk;
  double t = C.r.t0, h;
  double y = rowlive ? wm_init_value(model, (int64_t)n, (int64_t)ln, t, Pf) : 0.0;
  double f0 = rhs_of(y, t);
  bool lu_singular = false;
  TMP_DECL
  {
    const bool is_neg_h = C.r.h0 < 0.0;
    const double d0 = sqrt(wms_wave(y, y)), d1 = sqrt(wms_wave(f0, y));
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    const double hh = is_neg_h ? -h0 : h0;
    const double y1 = f0 * hh + y;
    const double f1 = rhs_of(y1, is_neg_h ? t - h0 : t + h0);
    const double df = f1 - f0;
    const double d2 = sqrt(wms_wave(df, y)) / fabs(h0);
    double max_d = d2;
    if (max_d < d1) max_d = d1;
    double h1;
    if (max_d < 1e-15) { h1 = h0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
    else h1 = rpow(0.01 / max_d, 1.0 / (1.0 + 1.0), det);
    h = 100.0 * h0;
    if (h > h1) h = h1;
    if (is_neg_h) h = -h;
  }

  // ------------------------------------------------------------ Bdf::_new
  int order = 1;
  double D[kNC], Dt[kNC];
#pragma unroll
  for (int j = 0; j < kNC; ++j) { D[j] = 0.0; Dt[j] = 0.0; }
  D[0] = y; D[1] = f0 * h;
  double opc = h * C.alpha[1];
  // ---- forward sensitivities: new_with_sensitivities_and_consistent (state.rs:1032-1083), new_augmented (bdf.rs:384-432): sdiff_j[:, 0] = s_j, sdiff_j[:, 1] = h ds_j
  constexpr int SP = SENS ? kWmMaxSensParams : 1, SC = SENS ? kNC : 1;
  double S[SP][SC], s_cur[SP], s_delta[SP];
  double s_c = 0.0;  // BdfCallable::c of the sensitivity operator: 0 until the first _update_step_size (op/bdf.rs:61)
  const int nsp = SENS ? C.np : 0;
  auto X2f = [&](int64_t k) { return xs2[k]; };
  // (1/n) sum_i (v_i / (|w_i| sens_rtol + sens_atol))^2, summed in index order
  auto wms_sens = [&](double v_mine, double w_mine) __attribute__((always_inline)) -> double {
    const double term = rowlive ? v_mine / (fabs(w_mine) * C.sens_rtol + C.sens_atol) : 0.0;
    __syncthreads();
    red[ln] = term * term;
    __syncthreads();
    return team_seq_sum(red, n) / (double)n;
  };
  if constexpr (SENS) {
    for (int j = 0; j < nsp; ++j) {
      auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
      __syncthreads();
      xs[ln] = y;
      __syncthreads();
      const double s0 = rowlive ? wm_sens_component(t, (int64_t)ln, Xf, Ej, Pf, true) : 0.0;
      const double dfdp = rowlive ? wm_sens_component(t, (int64_t)ln, Xf, Ej, Pf, false) : 0.0;  // SensRhs::update_state(y0, t0): column j of df/dp
      xs2[ln] = s0;
      __syncthreads();
      const double jm = rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, X2f, Pf, true) : 0.0;  // SensRhs::call_inplace: J(y0) s_j + (df/dp)_j
      const double ds = jm + dfdp;
#pragma unroll
      for (int k = 0; k < kNC; ++k) S[j][k] = 0.0;
      S[j][0] = s0; S[j][1] = ds * h;
      s_cur[j] = s0; s_delta[j] = 0.0;
    }
  }
  bool jac_stale = true;
  // The factorisation is by far the largest piece of code of this kernel: every request for a new linearisation only records what the reference
  // would have used (the value of c at that moment; state and time do not change before the next Newton solve) and the one inlined copy of
  // reset_jacobian runs at the top of the next solve attempt.
  bool reset_pending = true;
  double c_reset = opc;
  int n_setups = 0, n_steps = 0, n_err_fails = 0, n_newton = 0, n_nl_fails = 0;
  // reset_jacobian: J(x, t) row by row into LDS when stale, A = J * (-c) + I, LU in registers
  auto reset_jacobian = [&](double x_mine, double tt) __attribute__((always_inline)) {
    if constexpr (RL > 0) {
      if (jac_stale) {
        __syncthreads();
        xs[ln] = x_mine;
        __syncthreads();
        for (int jj = 0; jj < 64; ++jj) {  // my columns (the elimination's layout, trg_gcol; written and read by this thread only)
          const int j = trg_gcol(half, jj);
          auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
          if (rowlive && j < n) sJ[(size_t)j * n + ln] = wm_component(model, (int64_t)n, tt, (int64_t)ln, Xf, Ej, Pf, true);
        }
        jac_stale = false;
        TMP_MARK(0)
      }
#pragma unroll
      for (int jj = 0; jj < 64; ++jj) {
        const int j = trg_gcol(half, jj);
        a_rl[jj] = (rowlive && j < n) ? sJ[(size_t)j * n + ln] * (-c_reset) + (j == ln ? 1.0 : 0.0) : 0.0;
      }
      team_reg_lu_factor<RL>(a_rl, n, (int)threadIdx.x, wk, lu_singular, rl_d, rl_r);
      TMP_MARK(1)
      return;
    }
    if (jac_stale) {
      __syncthreads();
      xs[ln] = x_mine;
      __syncthreads();
      for (int j = 0; j < n; ++j) {
        auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
        if (rowlive) sJ[(size_t)j * n + ln] = wm_component(model, (int64_t)n, tt, (int64_t)ln, Xf, Ej, Pf, true);
      }
      jac_stale = false;
      TMP_MARK(0)
    }
    // A = J * (-c) + I (scale_add_and_assign with the dense identity mass), my row; then the factorisation in LDS
    if (rowlive)
      for (int j = 0; j < n; ++j) A[j * P + ln] = sJ[(size_t)j * n + ln] * (-c_reset) + (j == ln ? 1.0 : 0.0);
    team_lu_factor<W>(A, P, n, ln, rowlive, cand, perm, lu_singular);
    TMP_MARK(1)
  };
  auto lu_solve = [&](double& rhs) __attribute__((always_inline)) -> bool {
    if constexpr (RL > 0) return team_reg_lu_solve<RL>(a_rl, n, (int)threadIdx.x, wk, lu_singular, rl_d, rl_r, rhs);
    else return team_lu_solve<W>(A, P, n, ln, rowlive, perm, xs2, lu_singular, rhs);
  };
  n_setups = 1;
  // RootFinder::init
  double g0[2] = {1.0, 1.0};
  double rf_t0 = t;
  auto root_of = [&](double x_mine, double tt, double (&g)[2]) __attribute__((always_inline)) {
    __syncthreads();
    xs[ln] = x_mine;
    __syncthreads();
    g[0] = 1.0; g[1] = 1.0;  // unused slots: never zero, never a sign change
    double gg[2] = {0.0, 0.0};
    const int nr = wm_root_values(model, (int64_t)n, tt, Xf, Pf, gg);
    if (nr > 0) g[0] = gg[0];
    if (nr > 1) g[1] = gg[1];
  };
  if (C.nroots > 0) root_of(y, t, g0);
  double t_root = 0.0;
  int root_idx = -1;
  int steps_since_jac = 0, steps_since_rhs_jac = 0;
  double h_at_last_jac = 1.0;
  double eta = C.r.eta_reset;
  int n_equal_steps = 0;
  bool has_prev_err = false;
  double prev_err = 0.0;
  double yp = 0.0, psi = 0.0;
  double t_predict = t;

  auto update_step_size = [&](double factor, double& new_h_out) __attribute__((always_inline)) -> bool {
    TMP_MARK(7)
    const double new_h = factor * h;
    n_equal_steps = 0;
    double R[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      R[j][0] = 1.0;
#pragma unroll
      for (int i = 1; i < 6; ++i) R[j][i] = (j == 0) ? 0.0 : R[j][i - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
    }
    const double* U = C.u[order - 1];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j <= order) {
        double ru[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
          for (int m = 1; m < 6; ++m) if (m <= order) acc = R[m][k] * U[j * 6 + m] + acc;
          ru[k] = acc;
        }
        double acc = D[0] * ru[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) if (k <= order) acc = D[k] * ru[k] + acc;
        Dt[j] = acc;
      }
    }
#pragma unroll
    for (int j = 0; j < kNC; ++j) { const double tmp = D[j]; D[j] = Dt[j]; Dt[j] = tmp; }
    if constexpr (SENS) {
      // bdf.rs:546-548: every sdiff goes through the SAME scratch matrix as the states' differences, so the columns behind `order` are handed down the chain
      for (int q = 0; q < nsp; ++q) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (j <= order) {
            double ru[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
              for (int m = 1; m < 6; ++m) if (m <= order) acc = R[m][k] * U[j * 6 + m] + acc;
              ru[k] = acc;
            }
            double acc = S[q][0] * ru[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) if (k <= order) acc = S[q][k] * ru[k] + acc;
            Dt[j] = acc;
          }
        }
#pragma unroll
        for (int j = 0; j < kNC; ++j) { const double tmp = S[q][j]; S[q][j] = Dt[j]; Dt[j] = tmp; }
      }
      s_c = new_h * C.alpha[order];
    }
    opc = new_h * C.alpha[order];
    h = new_h;
    eta = C.r.eta_reset_ts;
    new_h_out = new_h;
    TMP_MARK(5)
    return fabs(h) < o.min_timestep;
  };
  auto predict_forward = [&]() __attribute__((always_inline)) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) if (j <= order) s = s + D[j];
    double q = C.gamma[1] * D[1];
#pragma unroll
    for (int j = 2; j < 6; ++j) if (j <= order) q = C.gamma[j] * D[j] + 1.0 * q;
    q = q * C.alpha[order];
    q = q - s;
    yp = s;
    psi = q;
    t_predict = t + h;
  };
  auto jacobian_updates = [&](double c, JState st) __attribute__((always_inline)) {
    bool check_rhs = false, check_jac = true;
    const double rel = fabs(c / h_at_last_jac - 1.0);
    switch (st) {
      case JState::StepSuccess:
        check_rhs = steps_since_rhs_jac >= o.update_rhs_jacobian_after_steps;
        check_jac = steps_since_jac >= o.update_jacobian_after_steps || rel > o.threshold_to_update_jacobian;
        break;
      case JState::FirstConvergenceFail: check_rhs = rel < o.threshold_to_update_rhs_jacobian; break;
      case JState::SecondConvergenceFail: check_rhs = steps_since_rhs_jac > 0; break;
      case JState::ErrorTestFail: check_rhs = false; break;
    }
    if (check_rhs) {
      jac_stale = true;
      reset_pending = true; c_reset = opc;
      steps_since_rhs_jac = 0; steps_since_jac = 0; h_at_last_jac = c;
      eta = C.r.eta_reset;
      n_setups++;
    } else if (check_jac) {
      reset_pending = true; c_reset = opc;
      steps_since_jac = 0; h_at_last_jac = c;
      eta = C.r.eta_reset;
      n_setups++;
    }
  };
  bool has_tstop = true;
  const double tstop = t_eval[C.r.n_eval - 1];
  auto handle_tstop = [&]() __attribute__((always_inline)) -> int {
    const double troundoff = 100.0 * kEps * (fabs(t) + fabs(h));
    if (fabs(t - tstop) <= troundoff) { has_tstop = false; return 1; }
    if ((h > 0.0 && tstop < t - troundoff) || (h < 0.0 && tstop > t + troundoff)) { has_tstop = false; return 2; }
    if ((h > 0.0 && t + h > tstop + troundoff) || (h < 0.0 && t + h < tstop - troundoff)) {
      const double factor = (tstop - t) / h;
      double nh;
      (void)update_step_size(factor, nh);
    }
    return 0;
  };
  auto interpolate = [&](double te) __attribute__((always_inline)) -> double {  // interpolate_from_diff, my component
    double time_factor = 1.0;
    double yv = D[0];
#pragma unroll
    for (int j = 0; j < kMaxOrder; ++j) {
      if (j < order) {
        const double jt = (double)j;
        time_factor *= (te - (t - h * jt)) / (h * (1.0 + jt));
        yv = time_factor * D[j + 1] + 1.0 * yv;
      }
    }
    return yv;
  };

  int col = 0;
  const bool steps_mode = !SENS && C.steps_cap > 0;  // every accepted step out (WaveMemberConsts::steps_cap)
  auto steps_write = [&](double tw, double yv_mine) __attribute__((always_inline)) {
    if (col < C.steps_cap) {
      if (lead) C.steps_t_out[(int64_t)col * nb + b] = tw;
      if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv_mine;
    }
    col++;
  };
  if (steps_mode) steps_write(t, y);  // write_out before the first step (method.rs:900)
  {
    const int r = handle_tstop();
    if (r == 1) status = kRsStopTimeAtCurrentTime;
    else if (r == 2) status = kRsStopTimeBeforeCurrentTime;
  }
  long guard = 0;
  bool done = status != kRsOk;
  while (!done) {
    if (++guard > o.max_steps) { status = kRsMaxStepsExceeded; break; }
    double safety = 0.0, error_norm = 0.0;
    const int old_err_fails = n_err_fails;
    bool convergence_fail = false;
    double x = 0.0;
    int niter = 0;
    predict_forward();
    while (true) {
      TMP_MARK(7)
      if (reset_pending) {
        if constexpr (RL > 0) {
          // The scalars of Bdf::step are the same in every thread but computed from sums read out of LDS: to the compiler they are per-lane values, and next to a
          // thread's 128 registers of M - c J there is no room for them (scratch traffic inside the elimination loops).  Read back from lane 0 they are scalars.
          auto sd = [](double& v) __attribute__((always_inline)) { v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); };
          auto si = [](int& v) __attribute__((always_inline)) { v = __builtin_amdgcn_readfirstlane(v); };
          sd(t); sd(h); sd(opc); sd(c_reset); sd(h_at_last_jac); sd(eta); sd(prev_err); sd(t_predict); sd(rf_t0); sd(t_root); sd(g0[0]); sd(g0[1]); sd(safety); sd(error_norm);
          si(order); si(n_setups); si(n_steps); si(n_err_fails); si(n_newton); si(n_nl_fails); si(steps_since_jac); si(steps_since_rhs_jac); si(n_equal_steps); si(col);
          si(status); si(root_idx); si(niter);
        }
        reset_jacobian(y, t);
        reset_pending = false;
      }
      x = yp;
      niter = 0;
      bool has_old = false;
      double old_norm = 0.0;
      bool solved = false;
      for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
        TMP_MARK(7)
        const double f = rhs_of(x, t_predict);
        TMP_MARK(2)
        const double tmpv = x + psi;
        double delta;
        if constexpr (kWmHasMass) {  // F(y) = M (y - y0 + psi) - c f(y): M's row times the published vector, then + (-c) f (mass_gemv with beta = -c)
          xs2[ln] = tmpv;
          __syncthreads();
          auto X2f = [&](int64_t k) { return xs2[k]; };
          delta = rowlive ? wm_mass_component(t_predict, (int64_t)ln, X2f, Pf) + (-opc) * f : 0.0;
          __syncthreads();
        } else {
          delta = 1.0 * tmpv + (-opc) * f;  // F(y) = (y - y0 + psi) - c f(y)
        }
        const bool lu_ok = lu_solve(delta);  // unknown i comes back to thread i
        TMP_MARK(3)
        if (!lu_ok) break;
        x = x - delta;
        const double norm = sqrt(wms_wave(delta, yp));
        TMP_MARK(4)
        niter += 1;
        bool diverged = false;
        if (has_old) {
          const double rate = niter == 2 ? norm / old_norm : rpow(norm / old_norm, 1.0 / (double)(niter - 1), det);
          if (rate > 0.9) diverged = true;
          else if (powi_rt(rate, o.max_nonlinear_solver_iterations - niter) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
          else eta = rate / (1.0 - rate);
        } else {
          const double min_eta = 1e4 * kEps;
          if (eta < min_eta) eta = min_eta;
          eta = rpow(eta, 0.8, det);
        }
        const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
        if (niter == 1) { has_old = true; old_norm = norm; }
        if (diverged) break;
        if (converged) { solved = true; break; }
      }
      n_newton += niter;
      if constexpr (SENS) {
        // sensitivity_solve (bdf.rs:934-989), as in k_bdf_wave_member: per parameter the predictor / psi of its difference array and a Newton solve of
        // F(s) = (s - s0 + psi) - c_s (J s + (df/dp)_j) with the factors of the state equations and the SHARED Convergence
        if (solved) {
          for (int j = 0; j < nsp && solved; ++j) {
            auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
            __syncthreads();
            xs[ln] = yp;
            __syncthreads();
            const double dfdp = rowlive ? wm_sens_component(t_predict, (int64_t)ln, Xf, Ej, Pf, false) : 0.0;
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k <= order) sacc = sacc + S[j][k];
            double q = C.gamma[1] * S[j][1];
#pragma unroll
            for (int k = 2; k < 6; ++k) if (k <= order) q = C.gamma[k] * S[j][k] + 1.0 * q;
            q = q * C.alpha[order];
            q = q - sacc;
            const double sp = sacc, spsi = q;
            double xsv = sacc;
            int sn = 0;
            bool s_has_old = false, s_solved = false;
            double s_old_norm = 0.0;
            for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
              __syncthreads();
              xs2[ln] = xsv;
              __syncthreads();
              const double jm = rowlive ? wm_component(model, (int64_t)n, t_predict, (int64_t)ln, Xf, X2f, Pf, true) : 0.0;
              __syncthreads();  // xs2 is the solve's exchange buffer next
              const double fr = jm + dfdp;
              double delta = 1.0 * (xsv + spsi) + (-s_c) * fr;
              const bool lu_ok = lu_solve(delta);
              if (!lu_ok) break;
              xsv = xsv - delta;
              const double norm = sqrt(wms_wave(delta, sp));
              sn += 1;
              bool diverged = false;
              if (s_has_old) {
                const double rate = sn == 2 ? norm / s_old_norm : rpow(norm / s_old_norm, 1.0 / (double)(sn - 1), det);
                if (rate > 0.9) diverged = true;
                else if (powi_rt(rate, o.max_nonlinear_solver_iterations - sn) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
                else eta = rate / (1.0 - rate);
              } else {
                const double min_eta = 1e4 * kEps;
                if (eta < min_eta) eta = min_eta;
                eta = rpow(eta, 0.8, det);
              }
              const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
              if (sn == 1) { s_has_old = true; s_old_norm = norm; }
              if (diverged) break;
              if (converged) { s_solved = true; break; }
            }
            niter = sn;  // Convergence::niter is the last solve's: the safety factor below reads it
            if (!s_solved) { solved = false; break; }  // `?` before the iteration count is added
            n_newton += sn;
            s_cur[j] = xsv; s_delta[j] = xsv - sp;
          }
        }
      }
      if (!solved) {
        n_nl_fails += 1;
        if (n_nl_fails > o.max_nonlinear_solver_failures) { status = kRsTooManyNonlinearSolverFailures; break; }
        has_prev_err = false;
        if (convergence_fail) {
          double new_h;
          if (update_step_size(0.3, new_h)) { status = kRsStepSizeTooSmall; break; }
          jacobian_updates(new_h * C.alpha[order], JState::SecondConvergenceFail);
          predict_forward();
        } else {
          jacobian_updates(h * C.alpha[order], JState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      const double ydelta = x - yp;
      error_norm = fmax(0.0, wms_wave(ydelta, y) * C.ec2[order - 1]);
      if constexpr (SENS) {
        if (C.sens_error_control)  // bdf.rs:844-858 — error_const2[order], not [order - 1]
          for (int j = 0; j < nsp; ++j) error_norm = fmax(error_norm, wms_sens(s_delta[j], s_cur[j]) * C.ec2[order]);
      }
      const double maxiter = (double)o.max_nonlinear_solver_iterations;
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + (double)niter);
      if (error_norm <= 1.0) {
        double dk1 = 0.0;
#pragma unroll
        for (int j = 2; j < 7; ++j) if (j == order + 1) dk1 = D[j];
        const double dk2 = ydelta - dk1;
#pragma unroll
        for (int j = 2; j < kNC; ++j) { if (j == order + 2) D[j] = dk2; if (j == order + 1) D[j] = ydelta; }
        double upper = ydelta;
#pragma unroll
        for (int j = 5; j >= 0; --j) if (j <= order) { const double v = D[j] + 1.0 * upper; D[j] = v; upper = v; }
        if constexpr (SENS) {  // update_differences_and_integrate_out (bdf.rs:628-643): _update_diff on every sensitivity difference array
          for (int q = 0; q < nsp; ++q) {
            const double sd = s_delta[q];
            double sk1 = 0.0;
#pragma unroll
            for (int j = 2; j < 7; ++j) if (j == order + 1) sk1 = S[q][j];
            const double sk2 = sd - sk1;
#pragma unroll
            for (int j = 2; j < kNC; ++j) { if (j == order + 2) S[q][j] = sk2; if (j == order + 1) S[q][j] = sd; }
            double up = sd;
#pragma unroll
            for (int j = 5; j >= 0; --j) if (j <= order) { const double v = S[q][j] + 1.0 * up; S[q][j] = v; up = v; }
          }
        }
        y = yp;
        t = t_predict;
        break;
      }
      double factor = safety * pi_controller_raw(error_norm, has_prev_err, prev_err, o.pi_control_integral, o.pi_control_proportional, order + 1, det);
      has_prev_err = false;
      if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
      double new_h;
      if (update_step_size(factor, new_h)) { status = kRsStepSizeTooSmall; break; }
      jacobian_updates(new_h * C.alpha[order], JState::ErrorTestFail);
      predict_forward();
      n_err_fails += 1;
      if (n_err_fails - old_err_fails >= o.max_error_test_failures) { status = kRsTooManyErrorTestFailures; break; }
    }
    if (status != kRsOk) break;
    n_steps += 1;
    steps_since_jac += 1; steps_since_rhs_jac += 1;
    prev_err = error_norm; has_prev_err = true;
    n_equal_steps += 1;
    if (n_equal_steps > order) {
      double vm = 0.0, vp = 0.0;
#pragma unroll
      for (int j = 1; j < kNC; ++j) { if (j == order) vm = D[j]; if (j == order + 2) vp = D[j]; }
      const double inf = __builtin_huge_val();
      double error_m_norm = order > 1 ? wms_wave(vm, y) * C.ec2[order - 1] : inf;
      double error_p_norm = order < kMaxOrder ? wms_wave(vp, y) * C.ec2[order + 1] : inf;
      if constexpr (SENS) {  // predict_error_control with the augmented system (bdf.rs:871-932): the `error_norm.max(err)` chain from zero, then the sensitivities' terms
        if (order > 1) error_m_norm = fmax(0.0, error_m_norm);
        if (order < kMaxOrder) error_p_norm = fmax(0.0, error_p_norm);
        if (C.sens_error_control)
          for (int q = 0; q < nsp; ++q) {
            double cm = 0.0, cp = 0.0;
#pragma unroll
            for (int j = 1; j < kNC; ++j) { if (j == order) cm = S[q][j]; if (j == order + 2) cp = S[q][j]; }
            if (order > 1) error_m_norm = fmax(error_m_norm, wms_sens(cm, s_cur[q]) * C.ec2[order - 1]);
            if (order < kMaxOrder) error_p_norm = fmax(error_p_norm, wms_sens(cp, s_cur[q]) * C.ec2[order + 1]);
          }
      }
      const double pi_i = o.pi_control_integral, pi_p = o.pi_control_proportional;
      const double f0c = pi_controller_raw(error_m_norm, has_prev_err, prev_err, pi_i, pi_p, order, det);
      const double f1c = pi_controller_raw(error_norm, has_prev_err, prev_err, pi_i, pi_p, order + 1, det);
      const double f2c = pi_controller_raw(error_p_norm, has_prev_err, prev_err, pi_i, pi_p, order + 2, det);
      int max_index = 0;
      double fmaxv = f0c;
      if (f1c >= fmaxv) { max_index = 1; fmaxv = f1c; }
      if (f2c >= fmaxv) { max_index = 2; fmaxv = f2c; }
      const int new_order = max_index == 0 ? order - 1 : (max_index == 1 ? order : order + 1);
      order = new_order;
      double factor = safety * fmaxv;
      if (factor > o.max_timestep_growth) factor = o.max_timestep_growth;
      if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
      if (factor >= o.min_timestep_growth || factor <= o.max_timestep_shrink || max_index == 0 || max_index == 2) {
        double new_h;
        if (update_step_size(factor, new_h)) { status = kRsStepSizeTooSmall; break; }
        jacobian_updates(new_h * C.alpha[new_order], JState::StepSuccess);
      }
    }
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if (C.nroots > 0) {
      // RootFinder::check_root (root.rs:91-222) on wavefront-uniform root values
      double g1[2], gmid[2];
      root_of(y, t, g1);
      bool found;
      double frac;
      int imax;
      root_finding_lane<2>(g0, g1, found, frac, imax);
      if (imax < 0) {
        g0[0] = g1[0]; g0[1] = g1[1];
        rf_t0 = t;
        if (found) { t_root = t; root_idx = fabs(g0[1]) < fabs(g0[0]) && C.nroots > 1 ? 1 : 0; reason = 3; }
      } else {
        double alpha = 1.0;
        bool sc0 = false, sc1 = true;
        int itr = 0;
        double t1 = t, t0l = rf_t0;
        const double tol = 100.0 * kEps * (fabs(t1) + fabs(t1 - t0l));
        bool early = false;
        while (fabs(t1 - t0l) > tol) {
          const double g1v = imax == 0 ? g1[0] : g1[1], g0v = imax == 0 ? g0[0] : g0[1];
          double t_mid = t1 - (t1 - t0l) * g1v / (g1v - alpha * g0v);
          if (fabs(t_mid - t0l) < 0.5 * tol) {
            const double fracint = fabs(t1 - t0l) / tol;
            const double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
            t_mid = t0l + fracsub * (t1 - t0l);
          }
          if (fabs(t1 - t_mid) < 0.5 * tol) {
            const double fracint = fabs(t1 - t0l) / tol;
            const double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
            t_mid = t1 - fracsub * (t1 - t0l);
          }
          root_of(interpolate(t_mid), t_mid, gmid);
          bool f2;
          double fr2;
          int i2;
          root_finding_lane<2>(g0, gmid, f2, fr2, i2);
          const bool lower = i2 >= 0;
          if (lower) {
            t1 = t_mid; imax = i2;
            g1[0] = gmid[0]; g1[1] = gmid[1];
          } else if (f2) {
            root_of(y, t, g0);
            t_root = t_mid; root_idx = imax; early = true;
            break;
          } else {
            t0l = t_mid;
            g0[0] = gmid[0]; g0[1] = gmid[1];
          }
          if ((itr & 1) == 0) sc0 = lower; else sc1 = lower;
          if (itr >= 2) alpha = (sc0 != sc1) ? 1.0 : (sc0 ? 0.5 * alpha : 2.0 * alpha);
          itr += 1;
        }
        if (!early) { root_of(y, t, g0); t_root = t1; root_idx = imax; }
        reason = 3;
      }
    }
    if (reason == 0 && has_tstop) reason = handle_tstop();
    if (reason == 2) reason = 0;
    const double upto = reason == 3 ? t_root : t;
    if (steps_mode) {  // InternalTimestep / TstopReached -> write_out (method.rs:907-921): state.y; a root is written below, at the root
      if (reason != 3) steps_write(t, y);
    } else
    while (col < C.r.n_eval && t_eval[col] <= upto) {
      const double yv = interpolate(t_eval[col]);
      if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv;
      if constexpr (SENS) {  // interpolate_sens (bdf.rs:1162-1215): the same polynomial on every sensitivity difference array
        const double te = t_eval[col];
        for (int q = 0; q < nsp; ++q) {
          double time_factor = 1.0;
          double sv = S[q][0];
#pragma unroll
          for (int j = 0; j < kMaxOrder; ++j) {
            if (j < order) {
              const double jt = (double)j;
              time_factor *= (te - (t - h * jt)) / (h * (1.0 + jt));
              sv = time_factor * S[q][j + 1] + 1.0 * sv;
            }
          }
          if (rowlive) C.sens_out[(((int64_t)col * nsp + q) * n + ln) * nb + b] = sv;
        }
      }
      col++;
    }
    if constexpr (kWmResets) {
      if (reason == 3) {
        // A reset operator is configured: as in k_bdf_wave_member — state back to the root, y <- reset(y, t), dy <- f(y, t), stop time armed again on the old differences
        // and order, restart at first order (method.rs:774-797; bdf.rs:1232-1262, :1017-1020, :1290-1318)
        const double yb = interpolate(t_root);
        t = t_root;
        __syncthreads();
        xs[ln] = yb;
        __syncthreads();
        y = rowlive ? wm_reset_component(t, (int64_t)ln, Xf, Pf) : 0.0;
        const double dyr = rhs_of(y, t);
        if (steps_mode) steps_write(t, y);  // method.rs:931-932: the reset state at the root time
        if (t < tstop) {
          has_tstop = true;
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          root_of(y, t, g0);
          rf_t0 = t;
          n_equal_steps = 0;
          order = 1;
          D[0] = y; D[1] = dyr * h;
          opc = h * C.alpha[1];
          jacobian_updates(h * C.alpha[1], JState::StepSuccess);
          has_prev_err = false;
          if (has_tstop) { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          reason = 0;
        } else {
          done = true;
          reason = 0;
        }
      }
    }
    if (reason == 3 && steps_mode) {  // method.rs:922-947 without a reset: state_mut_back(t_root), write_out, RootFound
      const double yv = interpolate(t_root);
      steps_write(t_root, yv);
      done = true;
    } else
    if (reason == 3) {
      if (col < C.r.n_eval) {
        const double yv = interpolate(t_root);
        if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv;
        col++;
      }
      done = true;
    }
    if (reason == 1) done = true;
  }
  TMP_MARK(7)
  TMP_PRINT
  const int ncols = col;
  if (!steps_mode)
  for (; col < C.r.n_eval; ++col) {
    if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = __builtin_nan("");
    if constexpr (SENS)
      for (int q = 0; q < nsp; ++q)
        if (rowlive) C.sens_out[(((int64_t)col * nsp + q) * n + ln) * nb + b] = __builtin_nan("");
  }
  if (lead) {
    if (ncols_out != nullptr) ncols_out[b] = ncols;
    if (t_root_out != nullptr) t_root_out[b] = root_idx >= 0 ? t_root : __builtin_nan("");
    if (root_idx_out != nullptr) root_idx_out[b] = root_idx;
    if (status_out != nullptr) status_out[b] = status;
    if (stats_out != nullptr) {
      stats_out[0 * nb + b] = n_steps;
      stats_out[1 * nb + b] = n_newton;
      stats_out[2 * nb + b] = n_setups;
      stats_out[3 * nb + b] = n_err_fails;
      stats_out[4 * nb + b] = n_nl_fails;
    }
    atomicAdd(&totals[0], (unsigned long long)n_steps);
    atomicAdd(&totals[1], (unsigned long long)n_newton);
    atomicAdd(&totals[2], (unsigned long long)n_setups);
    atomicAdd(&totals[3], (unsigned long long)n_err_fails);
    atomicAdd(&totals[4], (unsigned long long)n_nl_fails);
    if (status != kRsOk) atomicAdd(&totals[5], 1ull);
  }
}

template <int W, bool SENS = false>
__global__ __launch_bounds__(64 * W) void k_bdf_team_member(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, int atol_broadcast,
                                                       const WaveMemberConsts* __restrict__ Cp, const double* __restrict__ t_eval, double* __restrict__ jac_scratch, double* __restrict__ y_out,
                                                       int32_t* __restrict__ stats_out, int32_t* __restrict__ status_out, double* __restrict__ t_root_out,
                                                       int32_t* __restrict__ root_idx_out, int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  bdf_team_member_body<W, SENS, 0>(nb, p_g, atol_g, atol_broadcast, Cp, t_eval, jac_scratch, y_out, stats_out, status_out, t_root_out, root_idx_out, ncols_out, totals);
}
// n <= NL <= 128, the factors in registers: four wavefronts (NL <= 64: two), two per SIMD (256 registers a thread), two (four) members on a CU
template <int NL>
__global__ __launch_bounds__(trg_threads(NL), 2) void k_bdf_team_member_rl(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, int atol_broadcast,
                                                       const WaveMemberConsts* __restrict__ Cp, const double* __restrict__ t_eval, double* __restrict__ jac_scratch, double* __restrict__ y_out,
                                                       int32_t* __restrict__ stats_out, int32_t* __restrict__ status_out, double* __restrict__ t_root_out,
                                                       int32_t* __restrict__ root_idx_out, int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  bdf_team_member_body<2, false, NL>(nb, p_g, atol_g, atol_broadcast, Cp, t_eval, jac_scratch, y_out, stats_out, status_out, t_root_out, root_idx_out, ncols_out, totals);
}

}  // namespace dsh
