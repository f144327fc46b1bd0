// Banded solve for SMALL ensembles, third form: one WORKGROUP per 16 systems, its wavefronts specialised.
//
// k_lu_band_solve_wide (dsh_lu_band.hpp) gave a wavefront 8 systems and let all 64 lanes prefetch for the 8 that run the chain.  Its profile
// (profiles/r02_pmc_band_wide.json) says what is left: the wavefront is ONE instruction stream, so every staging store, every load issue and every result
// store of a chunk stands in the chain lanes' way, and the prefetch depth is what one wavefront's registers hold (~40 loads).  76-80 us at n = 512 x 4096,
// of which 61 us are the chain wavefront's own instruction stream.
//
// Here the roles are separate wavefronts of one workgroup (320 threads):
//   * wavefront 4, lanes 0..15 — the CHAIN: per step it reads its operands from LDS (all reads of a chunk issued up front), does the arithmetic of
//     k_lu_band_solve (same operations, same order: bit-identical solutions) and writes the solution entry to LDS.  Nothing else is in its stream.
//   * wavefronts 0..3 — the LOADERS: 256 lanes = 16 row groups x 16 systems.  A load instruction fetches one operand of 4 consecutive rows, 128 contiguous
//     bytes per row (16 neighbouring systems of the batch-fastest layout).  Each lane keeps RF chunks in flight in registers (up to ~48 loads per lane,
//     ~100 KB per workgroup), writes the chunk that has landed into the LDS slot the chain will read next, computes what can be computed off the chain
//     (pivot offsets, the denominator half of the division, whether the chunk can run the stripped step) and stores the previous chunk's results.
// One s_barrier per chunk hands the slots over (two slots: the chain reads one while the loaders fill the other).
// Layouts and arithmetic: see k_lu_band_solve / k_lu_band_solve_wide; the stripped / general variants of a step are theirs.
#pragma once
#include <type_traits>
#include <utility>

#include "dsh_lu_band.hpp"

namespace dsh {

constexpr int kTeamLoaders = 256;  // loader lanes (4 wavefronts)
constexpr int kTeamThreads = kTeamLoaders + 64;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) in turn: the stages of one unrolled trip with their index as a compile-time constant
template <class F, int... I>
__device__ __forceinline__ void team_for_each_stage(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }

// hands the LDS slots over: LDS traffic only — __syncthreads() would also wait for the loaders' loads in flight (vmcnt(0)) and drain the prefetch every chunk
__device__ __forceinline__ void team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SYS = systems per workgroup (16, 32 or 64: the chain wavefront's instruction stream costs the same whatever its number of active lanes, so the
// ensemble is spread over as many workgroups as there are CUs — 16 per workgroup up to 4096 systems)
template <int K, int SYS>
struct band_team_cfg {
  static constexpr int R = K + 1, C = 2 * K + 1;
  static constexpr int CH = (K == 1 ? 512 : 256) / SYS;  // steps per chunk
  static constexpr int Q = CH * SYS / kTeamLoaders;       // rows per loader lane and chunk
  static constexpr int FO = K + 2;                        // forward operands per step: K multipliers, pivot offset, the entry that enters the window
  static constexpr int BO = C + 1;                        // backward loads per step: C entries of U, the entry that enters the window
  static constexpr int MO = C + 2;                        // LDS operand planes: backward has one more, the denominator half of the division
  static constexpr int RF = (48 / (FO * Q)) < 2 ? 2 : ((48 / (FO * Q)) > 8 ? 8 : (48 / (FO * Q)));  // chunks in flight per lane
  static constexpr int RB = (48 / (BO * Q)) < 2 ? 2 : ((48 / (BO * Q)) > 8 ? 8 : (48 / (BO * Q)));
};

#ifdef DSH_TEAM_PROF
__device__ unsigned long long g_team_prof[8];
__device__ unsigned long long g_team_stamp[2][8];
#define TEAM_STAMP(k) if ((threadIdx.x == kTeamLoaders) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) g_team_stamp[blockIdx.x == 0 ? 0 : 1][k] = wall_clock64();  // [0] chain busy fwd, [1] chain total fwd, [2] chain busy bwd, [3] chain total bwd, [4..7] the same for loader wavefront 0
#define TEAM_PROF_DECL unsigned long long pb_ = 0, pt0_ = clock64(), pl_ = pt0_;
#define TEAM_PROF_BUSY_BEGIN pl_ = clock64();
#define TEAM_PROF_BUSY_END pb_ += clock64() - pl_;
#define TEAM_PROF_STORE(slot) if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { g_team_prof[slot] = pb_; g_team_prof[slot + 1] = clock64() - pt0_; }
#else
#define TEAM_STAMP(k)
#define TEAM_PROF_DECL
#define TEAM_PROF_BUSY_BEGIN
#define TEAM_PROF_BUSY_END
#define TEAM_PROF_STORE(slot)
#endif

// EPI (round 5): what the caller does with the solution next, in the shadow of the chain — the weighted mean-square norm of the solution (Vector::squared_norm against
// y / atol / rtol: the Newton iteration's convergence norm, the SDIRK error estimate) and, optionally, the Newton update xout = xin - solution (NoLineSearch::
// take_optimal_step).  The backward loaders fetch y, atol (and xin) of the rows they fetch U for; the lanes that store a chunk's solution form
// term = x / (|y| rtol + atol), keep term * term in LDS ([row][system], the whole vector: n x SYS doubles) and store the update; when the sweep is over, one lane per
// system adds the n squares in index order — the terms, the order of the additions and the division by n of k_squared_norm: the same bits — and the norm leaves in the
// launch's record (max over the systems) next to the zero-pivot count.  Replaces a second launch that read three to four vectors (23.7 us at 512 x 4096) by ~3 us of
// additions at the end of this one.
struct band_epi_args {
  const double* xin;   // nullptr: norm only
  double* xout;
  const double* y;     // n x nb, or n (by != 0)
  const double* atol;  // n x nb, or n (ba != 0)
  double rtol;
  int by, ba;
};
constexpr int kTeamEpiPlanes = 3;  // y, atol, xin of a chunk's rows
template <int K, int SYS>
constexpr size_t band_team_epi_lds_bytes(int64_t n) { return sizeof(double) * ((size_t)n * SYS); }

template <int K, int SYS, bool EPI = false>
__global__ __launch_bounds__(kTeamThreads) void k_lu_band_solve_team(int64_t n, int64_t nb, const double* __restrict__ fac, const int32_t* __restrict__ piv,
                                                                     double* __restrict__ rhs, unsigned long long* rec, unsigned int seq, band_epi_args ea = band_epi_args()) {
  using Cfg = band_team_cfg<K, SYS>;
  constexpr int S = SYS, R = Cfg::R, C = Cfg::C, CH = Cfg::CH, Q = Cfg::Q, FO = Cfg::FO, BO = Cfg::BO, MO = Cfg::MO, RF = Cfg::RF, RB = Cfg::RB;
  constexpr int G = kTeamLoaders / S;  // row groups
  constexpr int SBF = CH < (K == 1 ? 16 : 8) ? CH : (K == 1 ? 16 : 8), SBB = CH < (K == 1 ? 8 : 4) ? CH : (K == 1 ? 8 : 4);  // chain: steps whose operands are read from LDS together (forward / backward)
  static_assert(CH % SBF == 0 && CH % SBB == 0, "sub-blocks must divide the chunk");
  static_assert(CH % G == 0 && Q >= 1, "chunk rows must divide over the row groups");
  __shared__ double sOp[2][MO][CH][S];
#ifdef DSH_TEAM_X_W128
  __shared__ __attribute__((aligned(16))) double sOutT[2][S][CH + 2];
#define TEAM_OUT(sl, t, s) sOutT[sl][s][t]
#else
  __shared__ double sOut[2][CH][S];
#define TEAM_OUT(sl, t, s) sOut[sl][t][s]
#endif
  __shared__ int sMoved[2][4];  // forward: did any system of the workgroup interchange in the chunk (one word per loader wavefront)
  extern __shared__ double sDynTeam[];  // EPI: [n][S] squared terms of the norm
  double* const sSq = sDynTeam;
  TEAM_STAMP(0)
  const int tid = threadIdx.x;
  const bool loader = tid < kTeamLoaders;
  const int lane = tid & 63;
  const int s = loader ? tid % S : lane % S;  // system within the workgroup
  const int g = tid / S;                      // loader: row group
  const bool chain = !loader && lane < S;
  if (!loader) __builtin_amdgcn_s_setprio(3);  // the chain wavefront shares its SIMD with one loader wavefront: when both can issue, the chain goes first
  const int64_t b0 = (int64_t)blockIdx.x * S + s;
  const bool valid = b0 < nb;
  const int64_t b = valid ? b0 : nb - 1;  // lanes past the ensemble shadow the last system (no stores)
  const int nch = (int)((n + CH - 1) / CH);
  const int ni = (int)n;
  unsigned long long bad = 0ull;
  const uint32_t nb8 = (uint32_t)nb * 8u, b8 = (uint32_t)b * 8u, last8 = (uint32_t)(n - 1) * nb8 + b8;  // byte offsets (the caller guarantees n * nb < 2^28)
  auto ld_f64 = [](const double* base, uint32_t off) __attribute__((always_inline)) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off); };
  auto ld_i32 = [](const int32_t* base, uint32_t off) __attribute__((always_inline)) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + off); };
  auto st_f64 = [](double* base, uint32_t off, double v) __attribute__((always_inline)) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v; };

  // ================================================================ forward: interchanges interleaved with the unit-lower-triangular solve
  if (loader) {
    double pf[RF][FO][Q];
    int pp[RF][Q];
    const double* lbase[K];
#pragma unroll
    for (int r = 0; r < K; ++r) lbase[r] = fac + (int64_t)(C + r) * n * nb;
    uint32_t offq[Q];  // byte offset of row c*CH + q*G + g of this lane's system, for the chunk issued next
#pragma unroll
    for (int q = 0; q < Q; ++q) offq[q] = (uint32_t)(q * G + g) * nb8 + b8;
    auto issue = [&](double (&pd)[FO][Q], int (&pi)[Q]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const uint32_t o = min(offq[q], last8);  // clamped, unconditional loads; steps beyond n are skipped by the chain
#pragma unroll
        for (int r = 0; r < K; ++r) pd[r][q] = ld_f64(lbase[r], o);
        pi[q] = ld_i32(piv, o >> 1);
        pd[K + 1][q] = ld_f64(rhs, min(offq[q] + (uint32_t)(1 + K) * nb8, last8));
        offq[q] += (uint32_t)CH * nb8;
      }
    };
    auto land = [&](double (&pd)[FO][Q], int (&pi)[Q], int c) __attribute__((always_inline)) {  // chunk c, landed in registers -> its LDS slot
      const int sl = c & 1;
      bool moved = false;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int t = q * G + g;
#pragma unroll
        for (int o = 0; o < FO; ++o) if (o != K) sOp[sl][o][t][s] = pd[o][q];
        const int off = pi[q] - min(c * CH + t, ni - 1);  // pivot row - step
        sOp[sl][K][t][s] = (double)off;
        moved = moved | (off != 0);
      }
      const bool any = __builtin_amdgcn_ballot_w64(moved) != 0ull;
      if (lane == 0) sMoved[sl][tid >> 6] = any ? 1 : 0;
    };
    auto drain = [&](int c) __attribute__((always_inline)) {  // results of chunk c: LDS -> memory
#ifdef DSH_TEAM_X_WGLOBAL
      return;
#endif
      const int sl = c & 1;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int t = q * G + g, j = c * CH + t;
        if (valid && j < ni) st_f64(rhs, (uint32_t)j * nb8 + b8, TEAM_OUT(sl, t, s));
      }
    };
#pragma unroll
    for (int d = 0; d < RF; ++d) issue(pf[d], pp[d]);
    land(pf[0], pp[0], 0);
    issue(pf[0], pp[0]);
    team_barrier();
    // phase c: the chain runs chunk c; the loaders store the results of chunk c-1 and hand chunk c+1 over.  Every trip runs all RF stages and every stage
    // issues its loads (clamped rows past the end): a fixed number of younger loads behind each landing keeps the compiler's s_waitcnt from draining the queue
    TEAM_PROF_DECL
    for (int c0 = 0; c0 < nch; c0 += RF) {
#pragma unroll
      for (int d = 0; d < RF; ++d) {
        const int c = c0 + d;
        TEAM_PROF_BUSY_BEGIN
        if (c >= 1 && c - 1 < nch) drain(c - 1);
        const int nx = (d + 1) % RF;  // compile-time after unrolling
        land(pf[nx], pp[nx], c + 1);
        issue(pf[nx], pp[nx]);
        TEAM_PROF_BUSY_END
        team_barrier();
      }
    }
    if (tid < 64) { TEAM_PROF_STORE(4) }
    if (((nch + RF - 1) / RF) * RF == nch) drain(nch - 1);  // otherwise the trips past the end have stored it
  } else {
    double v[R];
    if (chain) {
#pragma unroll
      for (int r = 0; r < R; ++r) { const double t = rhs[min((int64_t)r, n - 1) * nb + b]; v[r] = r < n ? t : 0.0; }
    }
    team_barrier();
    TEAM_STAMP(1)
    TEAM_PROF_DECL
    for (int c0 = 0; c0 < nch; c0 += RF) {
#pragma unroll
      for (int d = 0; d < RF; ++d) {
        const int c = c0 + d;
        const int sl = c & 1;
        TEAM_PROF_BUSY_BEGIN
        if (chain && c < nch) {
          const bool stripped = (c + 1) * CH + K < ni && (sMoved[sl][0] | sMoved[sl][1] | sMoved[sl][2] | sMoved[sl][3]) == 0;  // workgroup-uniform
          if (stripped) {
            // operands of SBF steps at a time, the next sub-block's LDS reads issued before the current one's arithmetic
            double l[2][SBF][K], nxt[2][SBF];
            auto fetch = [&](int sb, double (&ll)[SBF][K], double (&nn)[SBF]) __attribute__((always_inline)) {
#pragma unroll
              for (int t = 0; t < SBF; ++t) {
#pragma unroll
#ifdef DSH_TEAM_X_NOREAD
                for (int r = 0; r < K; ++r) ll[t][r] = -0.27;
                nn[t] = 0.3;
#else
                for (int r = 0; r < K; ++r) ll[t][r] = sOp[sl][r][sb * SBF + t][s];
                nn[t] = sOp[sl][K + 1][sb * SBF + t][s];
#endif
              }
            };
            fetch(0, l[0], nxt[0]);
#ifdef DSH_TEAM_X_SCHED
            __builtin_amdgcn_sched_group_barrier(0x100, SBF * (K + 1) / 2, 0);  // the first sub-block's reads first: every later read group then fetches a sub-block ahead
#endif
            double xprev = 0.0; (void)xprev;
#pragma unroll
            for (int sb = 0; sb < CH / SBF; ++sb) {
              if (sb + 1 < CH / SBF) fetch(sb + 1, l[(sb + 1) & 1], nxt[(sb + 1) & 1]);
#pragma unroll
              for (int t = 0; t < SBF; ++t) {
                const double x = v[0];
#if defined(DSH_TEAM_X_WGLOBAL)
                if (valid) st_f64(rhs, (uint32_t)(c * CH + sb * SBF + t) * nb8 + b8, x);
#elif defined(DSH_TEAM_X_W128)
                if (t & 1) { typedef double d2 __attribute__((ext_vector_type(2))); d2 pr = {xprev, x}; *reinterpret_cast<d2*>(&sOutT[sl][s][sb * SBF + t - 1]) = pr; } else xprev = x;
#elif !defined(DSH_TEAM_X_NOWRITE)
                TEAM_OUT(sl, sb * SBF + t, s) = x;
#endif
#pragma unroll
                for (int r = 1; r < R; ++r) v[r] = (-x) * l[sb & 1][t][r - 1] + v[r];
#pragma unroll
                for (int r = 0; r + 1 < R; ++r) v[r] = v[r + 1];
                v[R - 1] = nxt[sb & 1][t];
#ifdef DSH_TEAM_X_SCHED
                // the LDS traffic of the chain goes into the shadow of its dependent arithmetic: multiply, one LDS instruction, add, (store of the previous pair)
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
                if (t & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
#endif
              }
            }
          } else {
            const int jb = c * CH;
#pragma unroll 4
            for (int t = 0; t < CH; ++t) {
              const int j = jb + t;
              if (j < ni) {
                const int pv = (int)sOp[sl][K][t][s];
                const double top = v[0];
                double x = top;
#pragma unroll
                for (int r = 1; r < R; ++r) {
                  const bool sel = (r == pv);
                  const double cur = v[r];
                  x = sel ? cur : x;
                  v[r] = sel ? top : cur;
                }
                TEAM_OUT(sl, t, s) = x;
#pragma unroll
                for (int r = 1; r < R; ++r) v[r] = (-x) * sOp[sl][r - 1][t][s] + v[r];
#pragma unroll
                for (int r = 0; r + 1 < R; ++r) v[r] = v[r + 1];
                v[R - 1] = j + 1 + K < ni ? sOp[sl][K + 1][t][s] : 0.0;
              }
            }
          }
        }
        TEAM_PROF_BUSY_END
        team_barrier();
      }
    }
    TEAM_PROF_STORE(0)
    TEAM_STAMP(2)
  }
  // the backward sweep reads, from other wavefronts of the workgroup, what the forward sweep stored: workgroup scope (one CU, one L1) — an agent-scope
  // __threadfence() here costs an L2 write-back and invalidate per workgroup, 20 us at 512 x 4096
  __syncthreads();

  // ================================================================ backward with U (bandwidth 2K): chunk c covers rows n-1 - (c*CH + t)
  if (loader) {
    constexpr int BE = BO + (EPI ? kTeamEpiPlanes : 0);
    double pb[RB][BE][Q];
    const double* const xin_or_rhs = (EPI && ea.xin != nullptr) ? ea.xin : rhs;
    const bool sub = EPI && ea.xin != nullptr;
    int rowq[Q];  // EPI: row i of the chunk issued next (negative past the top of the matrix)
#pragma unroll
    for (int q = 0; q < Q; ++q) rowq[q] = ni - 1 - (q * G + g);
    // U(i-d, i) lives at fac[(d*n + i-d)*nb + b] = (fac + (d*n - d)*nb)[i*nb + b]: one offset per row serves every diagonal; a row above the matrix
    // (i - d < 0, its value is never used) is clamped to row 0 of its diagonal by a single max
    const double* ubase[C];
    int ulo[C];
#pragma unroll
    for (int d = 0; d < C; ++d) { ubase[d] = fac + ((int64_t)d * n - d) * nb; ulo[d] = (int)((uint32_t)d * nb8 + b8); }
    int offi[Q];  // byte offset of row i = n-1 - (c*CH + q*G + g), for the chunk issued next; negative past the top of the matrix
#pragma unroll
    for (int q = 0; q < Q; ++q) offi[q] = (ni - 1 - (q * G + g)) * (int)nb8 + (int)b8;
    auto issue = [&](double (&pd)[BE][Q]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int d = 0; d < C; ++d) pd[d][q] = ld_f64(ubase[d], (uint32_t)max(offi[q], ulo[d]));
        pd[C][q] = ld_f64(rhs, (uint32_t)max(offi[q] - C * (int)nb8, (int)b8));
        if constexpr (EPI) {  // the epilogue's operands of row i itself (clamped like the others; rows above the matrix are never used)
          // no branches here: a join between the loads would make the compiler wait for every load in flight (the prefetch queue is the point of this wavefront).
          // Broadcast operands differ in the offset only; without a Newton update xin is the right-hand side itself (loaded, never used)
          const uint32_t oi = (uint32_t)max(offi[q], (int)b8);
          const uint32_t orow = (uint32_t)max(rowq[q], 0) * 8u;
          pd[BO][q] = ld_f64(ea.y, ea.by ? orow : oi);
          pd[BO + 1][q] = ld_f64(ea.atol, ea.ba ? orow : oi);
          pd[BO + 2][q] = ld_f64(xin_or_rhs, oi);
          rowq[q] -= CH;
        }
        offi[q] -= CH * (int)nb8;
      }
    };
    // EPI: the epilogue's operands of a chunk stay in this lane's registers from the landing to the store of the chunk's results two steps later (the same lane does
    // both): two sets in turn, the set index a compile-time constant (RB is even, so the parity of a chunk is the parity of its stage in the unrolled trip)
    static_assert(!EPI || RB % 2 == 0, "the epilogue's register sets alternate with the stages of a trip");
    double ek[2][EPI ? kTeamEpiPlanes : 1][Q];
    auto land = [&](double (&pd)[BE][Q], int c, auto par) __attribute__((always_inline)) {
      const int sl = c & 1;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int t = q * G + g;
        if constexpr (EPI) {
#pragma unroll
          for (int o = 0; o < kTeamEpiPlanes; ++o) ek[decltype(par)::value][o][q] = pd[BO + o][q];
        }
#pragma unroll
        for (int o = 0; o < BO; ++o) sOp[sl][o][t][s] = pd[o][q];
        // the denominator half of the division, off the chain; a diagonal the split division cannot vouch for hands the chain a NaN: its quotient then fails
        // the chain's own range check and the chunk is run again the ordinary way
        sOp[sl][C + 1][t][s] = div_den_ok(pd[0][q]) ? div_refined_rcp(pd[0][q]) : __builtin_nan("");
      }
    };
    auto drain = [&](int c, auto par) __attribute__((always_inline)) {
#ifdef DSH_TEAM_X_WGLOBAL
      return;
#endif
      const int sl = c & 1;
      constexpr int PAR = decltype(par)::value;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int t = q * G + g, i = ni - 1 - (c * CH + t);
        const double xs = TEAM_OUT(sl, t, s);
        if (valid && i >= 0) st_f64(rhs, (uint32_t)i * nb8 + b8, xs);
        if constexpr (EPI) {
          if (i >= 0) {
            const double term = xs / (fabs(ek[PAR][0][q]) * ea.rtol + ek[PAR][1][q]);  // Vector::squared_norm's term (nalgebra_serial.rs:395-408)
            sSq[i * S + s] = term * term;
            if (sub & valid) st_f64(ea.xout, (uint32_t)i * nb8 + b8, ek[PAR][2][q] - xs);  // xn -= delta (line_search.rs:57-68)
          }
        }
      }
    };
#pragma unroll
    for (int d = 0; d < RB; ++d) issue(pb[d]);
    land(pb[0], 0, std::integral_constant<int, 0>());
    issue(pb[0]);
    team_barrier();
    TEAM_PROF_DECL
    for (int c0 = 0; c0 < nch; c0 += RB) {
      // one stage of a trip: chunk c = c0 + D is run by the chain; the loaders store chunk c - 1 and hand chunk c + 1 over (both of parity (D + 1) & 1)
      auto stage = [&](auto dd) __attribute__((always_inline)) {
        constexpr int D = decltype(dd)::value;
        const int c = c0 + D;
        TEAM_PROF_BUSY_BEGIN
        if (c >= 1 && c - 1 < nch) drain(c - 1, std::integral_constant<int, (D + 1) & 1>());
        constexpr int nx = (D + 1) % RB;
        land(pb[nx], c + 1, std::integral_constant<int, (D + 1) & 1>());
        issue(pb[nx]);
        TEAM_PROF_BUSY_END
        team_barrier();
      };
      team_for_each_stage(stage, std::make_integer_sequence<int, RB>());
    }
    if (tid < 64) { TEAM_PROF_STORE(6) }
    if (((nch + RB - 1) / RB) * RB == nch) {
      if ((nch - 1) & 1) drain(nch - 1, std::integral_constant<int, 1>()); else drain(nch - 1, std::integral_constant<int, 0>());
    }
  } else {
    double w[C];
    if (chain) {
#pragma unroll
      for (int q = 0; q < C; ++q) { const int64_t r = n - 1 - (C - 1) + q; const double t = rhs[max(r, (int64_t)0) * nb + b]; w[q] = r >= 0 ? t : 0.0; }
    }
    team_barrier();
    TEAM_STAMP(3)
    TEAM_PROF_DECL
    for (int c0 = 0; c0 < nch; c0 += RB) {
#pragma unroll
      for (int d = 0; d < RB; ++d) {
        const int c = c0 + d;
        const int sl = c & 1;
        TEAM_PROF_BUSY_BEGIN
        if (chain && c < nch) {
          // the general step; also the second run of a stripped chunk whose quotients could not be vouched for
          auto general = [&]() __attribute__((always_inline)) {
            const int ib = ni - 1 - c * CH;
#pragma unroll 4
            for (int t = 0; t < CH; ++t) {
              const int i = ib - t;
              if (i >= 0) {
                const double diag = sOp[sl][0][t][s];
                if (diag == 0.0) bad = 1ull;
                const double x = w[C - 1] / diag;
                TEAM_OUT(sl, t, s) = x;
#pragma unroll
                for (int d2 = 1; d2 < C; ++d2) { const double uu = i - d2 >= 0 ? sOp[sl][d2][t][s] : 0.0; w[C - 1 - d2] = (-x) * uu + w[C - 1 - d2]; }  // as k_lu_band_solve: entries above row 0 are zeros
#pragma unroll
                for (int q = C - 1; q > 0; --q) w[q] = w[q - 1];
                w[0] = i - C >= 0 ? sOp[sl][C][t][s] : 0.0;
              }
            }
          };
          const bool stripped = ni - 1 - (c * CH + CH - 1) - C >= 0;  // every row of the chunk has its whole band and a successor entering the window
          if (stripped) {
            double u[2][SBB][C], nxt[2][SBB], iv[2][SBB], w0[C];
            auto fetch = [&](int sb, double (&uu)[SBB][C], double (&nn)[SBB], double (&ii)[SBB]) __attribute__((always_inline)) {
#pragma unroll
              for (int t = 0; t < SBB; ++t) {
#pragma unroll
#ifdef DSH_TEAM_X_NOREAD
                for (int d2 = 0; d2 < C; ++d2) uu[t][d2] = 3.7 - d2;
                nn[t] = 0.3;
                ii[t] = 0.27;
#else
                for (int d2 = 0; d2 < C; ++d2) uu[t][d2] = sOp[sl][d2][sb * SBB + t][s];
                nn[t] = sOp[sl][C][sb * SBB + t][s];
                ii[t] = sOp[sl][C + 1][sb * SBB + t][s];
#endif
              }
            };
            fetch(0, u[0], nxt[0], iv[0]);
#ifdef DSH_TEAM_X_SCHED
            __builtin_amdgcn_sched_group_barrier(0x100, SBB * (C + 2) / 2, 0);
#endif
            double xprev = 0.0; (void)xprev;
#pragma unroll
            for (int q = 0; q < C; ++q) w0[q] = w[q];
            bool ok = true;
            constexpr unsigned kLo2 = (unsigned)(1023 - 250) << 21, kRange2 = (unsigned)500 << 21;
            unsigned okacc = 0u; (void)okacc; (void)kRange2;
#pragma unroll
            for (int sb = 0; sb < CH / SBB; ++sb) {
              if (sb + 1 < CH / SBB) fetch(sb + 1, u[(sb + 1) & 1], nxt[(sb + 1) & 1], iv[(sb + 1) & 1]);
#pragma unroll
              for (int t = 0; t < SBB; ++t) {
                const double x = div_by_refined(w[C - 1], u[sb & 1][t][0], iv[sb & 1][t]);
#if defined(DSH_TEAM_X_INTCHECK)
                okacc = max(okacc, ((unsigned)__double2hiint(x) << 1) - kLo2);  // exponent of |x| within [1023 - 250, 1023 + 250): one shift-add and one max, no compare on the chain
#elif !defined(DSH_TEAM_X_NOCHECK)
                ok = ok & div_quot_ok(x);  // were the quotients quotients?  (dsh_device.hpp; together with the loaders' check of the diagonal)
#endif
#if defined(DSH_TEAM_X_WGLOBAL)
                if (valid) st_f64(rhs, (uint32_t)(ni - 1 - (c * CH + sb * SBB + t)) * nb8 + b8, x);
#elif defined(DSH_TEAM_X_W128)
                if (t & 1) { typedef double d2 __attribute__((ext_vector_type(2))); d2 pr = {xprev, x}; *reinterpret_cast<d2*>(&sOutT[sl][s][sb * SBB + t - 1]) = pr; } else xprev = x;
#elif !defined(DSH_TEAM_X_NOWRITE)
                TEAM_OUT(sl, sb * SBB + t, s) = x;
#endif
#pragma unroll
                for (int d2 = 1; d2 < C; ++d2) w[C - 1 - d2] = (-x) * u[sb & 1][t][d2] + w[C - 1 - d2];
#pragma unroll
                for (int q = C - 1; q > 0; --q) w[q] = w[q - 1];
                w[0] = nxt[sb & 1][t];
#ifdef DSH_TEAM_X_SCHED
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // q = w * r
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // e = fma(-y, q, w)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // x = fma(e, r, q)
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // the two products
                if (t & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // the two sums, the range check
#endif
              }
            }
#ifdef DSH_TEAM_X_INTCHECK
            ok = okacc < kRange2;
#endif
            if (!ok) {  // zero / tiny / huge entries (rare): this system runs the chunk again from its saved window
#pragma unroll
              for (int q = 0; q < C; ++q) w[q] = w0[q];
              general();
            }
          } else {
            general();
          }
        }
        TEAM_PROF_BUSY_END
        team_barrier();
      }
    }
    TEAM_PROF_STORE(2)
    TEAM_STAMP(4)
  }
  unsigned long long nbits = 0ull;
  if constexpr (EPI) {
    __syncthreads();  // every chunk's squares are in LDS
    if (tid < S) {    // one lane per system: the n additions in index order, sixteen LDS reads in flight at a time
      double acc = 0.0;
      int i = 0;
      if (ni >= 32) {  // the additions are one dependent chain (~8 cycles each): two register sets of sixteen squares in turn, each refilled from LDS while the other is added
        double va[16], vb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) va[k] = sSq[k * S + s];
        for (; i + 32 <= ni; i += 32) {
#pragma unroll
          for (int k = 0; k < 16; ++k) vb[k] = sSq[(i + 16 + k) * S + s];
#pragma unroll
          for (int k = 0; k < 16; ++k) acc += va[k];
          const int nx = i + 32 + 16 <= ni ? i + 32 : 0;  // the set after next (re-reads rows 0..15 when there is none: never added)
#pragma unroll
          for (int k = 0; k < 16; ++k) va[k] = sSq[(nx + k) * S + s];
#pragma unroll
          for (int k = 0; k < 16; ++k) acc += vb[k];
        }
        if (i + 16 <= ni) {  // va holds rows i .. i+15
#pragma unroll
          for (int k = 0; k < 16; ++k) acc += va[k];
          i += 16;
        }
      }
      for (; i < ni; ++i) acc += sSq[i * S + s];
      if (valid) nbits = d2u(acc / (double)n);
    }
  }
  block_publish(nbits, 0ull, (chain && valid) ? bad : 0ull, rec, seq);
}

}  // namespace dsh
