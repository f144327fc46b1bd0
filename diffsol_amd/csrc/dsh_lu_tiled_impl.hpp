// The layout-dependent part of the matrix-core dense LU (included by dsh_lu_tiled.hpp once per layout, inside namespace dsh::tl_one / dsh::tl_two, with
// DSH_TL_LAYOUT8 = 0 / 1).  No include guard on purpose.
// RS = rows per lane of the panel's register layout (8: n <= 512, 16: n <= 1024); kWaves = wavefronts per workgroup (a workgroup per system); kSW = width
// of a sub-panel, the columns that live in registers during the pivot steps: wavefront w holds its columns w, w + kWaves, ... — four columns each in
// both layouts.
//   n <= 1024: one workgroup of eight wavefronts per CU, sub-panels of 32 columns.
//   n <= 512:  TWO workgroups of four wavefronts per CU (256 registers per lane and at most 80 KB of LDS each), sub-panels of 16 columns: one system's
//              pivot steps — a latency chain of ~1 us per step with the matrix cores idle — run under the other system's trailing update, and two such
//              chains interleave on a SIMD at almost no cost.
// Two workgroups per CU with 32-column sub-panels were measured first (profiles/r04_lu_bench.md, 512 x 4096): eight wavefronts of 128 registers 40.4 ms,
// four wavefronts holding eight columns each 51.0 ms, against 27.4 ms with one workgroup per CU — the register columns leave the pivot-step loop no room
// and it spills.  -DDSH_TL_LAYOUT8=0 keeps the one-workgroup layout for n <= 512 as well.
#ifndef DSH_TL_RT8
#define DSH_TL_RT8 3
#define DSH_TL_PF8 2
#endif
// Trailing phase: DSH_TL_COLS = 1 deals the COLUMNS to the wavefronts (tl_trailing_cols: no U12 chunk in LDS, so it does not care that a workgroup of the two-per-CU
// layout has 80 KB), 0 keeps the chunked form (tl_trailing).  Measured (profiles/r05_lu_bench.md): the column form makes the two-workgroup layout the faster one at
// n = 512 (25.4 against 27.4 ms for 4096 systems) and loses 4 % in the one-workgroup layout, so each layout gets its own.  DSH_TL_CT: column tiles per wavefront
// (4 spills: 16 doubles of U12 per tile stay in registers next to the operands one row tile ahead).
// DSH_TL_FUSED_STAGE: the later sub-panels of a panel take the earlier ones' eliminations on the way into the stage instead of through W (tl_panel).  Pays in the
// two-workgroup layout (four sub-panels of 16: 24.1 -> 23.3 ms at 512 x 4096), costs with 16 row slots per lane (962 x 256: 8.6 -> 9.0 ms): per layout.
#ifndef DSH_TL_STAGE_TU
#define DSH_TL_STAGE_TU 1
#endif
#ifdef DSH_TL_FUSED_STAGE_FORCE
#define DSH_TL_FUSED_STAGE DSH_TL_FUSED_STAGE_FORCE
#else
#define DSH_TL_FUSED_STAGE DSH_TL_LAYOUT8
#endif
#ifndef DSH_TL_FINISH_T
#define DSH_TL_FINISH_T 1
#endif
#ifndef DSH_TL_CT
#define DSH_TL_CT 3
#endif
#ifdef DSH_TL_COLS_FORCE
#define DSH_TL_COLS DSH_TL_COLS_FORCE
#else
#define DSH_TL_COLS DSH_TL_LAYOUT8
#endif
#ifndef DSH_TL_RT16
#define DSH_TL_RT16 3
#define DSH_TL_PF16 2
#endif
template <int RS> struct tl_cfg;
template <> struct tl_cfg<8> {
  static constexpr int kWaves = DSH_TL_LAYOUT8 ? 4 : 8;
  static constexpr int kSW = DSH_TL_LAYOUT8 ? 16 : 32;
  static constexpr int kMaxN = 512;                          // rows
  static constexpr int kLDP = DSH_TL_LAYOUT8 ? 80 : 208;     // pitch of the U12 chunk in LDS (doubles); LDP % 32 == 16 keeps the operand reads conflict-free
  static constexpr int kCH = DSH_TL_LAYOUT8 ? 64 : 208;      // columns per chunk (tiles of 16; U12 is one tile per wavefront at a time)
  static constexpr int kRT = DSH_TL_RT8;                     // row tiles per wavefront whose L21 operand stays in registers
  static constexpr int kPF = DSH_TL_PF8;                     // column tiles of C in flight ahead of the one being multiplied
  static constexpr int kCT = DSH_TL_CT;                      // column tiles per wavefront of the column-dealt trailing phase
};
template <> struct tl_cfg<16> {
  static constexpr int kWaves = 8;
  static constexpr int kSW = 32;
  static constexpr int kMaxN = 1024;
  static constexpr int kLDP = 208;
  static constexpr int kCH = 208;      // 13 tiles of 16
  static constexpr int kRT = DSH_TL_RT16;
  static constexpr int kPF = DSH_TL_PF16;
  static constexpr int kCT = DSH_TL_CT;
};
inline int tiled_threads(int64_t n) { return 64 * (n <= 512 ? tl_cfg<8>::kWaves : tl_cfg<16>::kWaves); }
// dynamic LDS (doubles): [0, 16 P) Lbuf during the pivot steps (P = 64 RS rows) / the operands of the matrix-core phases otherwise (panel: L11A, U'; trailing:
// u12s [64][LDP], l11 [48][49]); then Ubuf [SW][SW + 1]; then invd [4][16][17], the inverses of the panel's diagonal blocks, which live from the
// sub-panel that produces them to the end of the trailing phase
template <int RS> constexpr int tl_ubuf() { return kTlLC * 64 * RS; }  // offset of Ubuf
template <int RS> constexpr int tl_invd() {  // behind both Ubuf and the trailing phase's operands
  constexpr int a = tl_ubuf<RS>() + tl_cfg<RS>::kSW * (tl_cfg<RS>::kSW + 1), b = 64 * tl_cfg<RS>::kLDP + 48 * kTlL11P;
  return a > b ? a : b;
}
template <int RS> constexpr size_t tiled_lds_doubles() { return (size_t)tl_invd<RS>() + 4 * 16 * 17; }
inline size_t tiled_lds_bytes(int64_t n) { return sizeof(double) * (n <= 512 ? tiled_lds_doubles<8>() : tiled_lds_doubles<16>()); }

// Pivot search of step k by the wavefront that owns its column (register column JC), and everything that has to be published for it: the multipliers
// (column kk of Lbuf), {lane, slot} of the pivot row, the interchange bookkeeping.  Finished rows need no mask: the pivot row's own "multiplier" is
// published as 1, so the elimination of its step leaves an exact 0 (u - u * 1) in every column behind it, rows finished in earlier sub-panels and rows
// beyond n enter the stage as zeros, and a zero is no candidate unless the whole column is zero (the position scan below).
// Their multipliers in later steps are 0 * (1 / pivot) = 0, so nothing ever changes them again.  The search is a maximum of magnitudes — v_max ignores NaNs, as the sequential scan does.
// Positions (LDS) are looked at only when they decide: several rows of the largest magnitude, or a column without a positive entry.
template <int RS>
__device__ __forceinline__ void tl_co_search(const tl_col<RS>& col, int k, int cb, int pbase, double* __restrict__ dyn, short* s_pos, short* s_rowat, int* s_prow,
                                             int* s_ipiv, int* s_hdr, int* s_flags, int lane) {
  constexpr int P = 64 * RS;
  const int g = cb + k;
  const int kk = k & (kTlLC - 1);
  double* const lcol = dyn + kk * P + lane;
#ifdef TL_X_STEPPROF
  const unsigned long long ts0_ = __builtin_readcyclecounter();
#endif
  const int rg = s_rowat[g];  // the row at the diagonal position (used at the end: its latency hides behind the search)
  double mt[RS];
#pragma unroll
  for (int s = 0; s < RS; ++s) mt[s] = __builtin_fabs(col[s]);
#pragma unroll
  for (int w = RS / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int s = 0; s < w; ++s) mt[s] = __builtin_fmax(mt[s], mt[s + w]);
  double m = __builtin_fmax(mt[0], -1.0);  // -1: no number in the column
  m = tl_dpp_max<kDppQuadXor1>(m);
  m = tl_dpp_max<kDppQuadXor2>(m);
  m = tl_dpp_max<kDppRowHalfMirror>(m);
  m = tl_dpp_max<kDppRowMirror>(m);
  const double wm = __builtin_fmax(__builtin_fmax(tl_readlane_f64(m, 0), tl_readlane_f64(m, 16)), __builtin_fmax(tl_readlane_f64(m, 32), tl_readlane_f64(m, 48)));
  unsigned long long M[RS];
  int cnt = 0;
#pragma unroll
  for (int s = 0; s < RS; ++s) { M[s] = __builtin_amdgcn_ballot_w64(__builtin_fabs(col[s]) == wm); cnt += __popcll(M[s]); }
  int ls = 0, ss = 0;
  double piv = 0.0;
  if (cnt == 1 && wm > 0.0) {  // the common case: one row holds the largest magnitude
#pragma unroll
    for (int s = 0; s < RS; ++s)
      if (M[s] != 0ull) { ss = s; ls = __ffsll((long long)M[s]) - 1; piv = tl_readlane_f64(col[s], ls); }
  } else {
    // the smallest position among the candidates: the rows of the largest magnitude, or — no number in the column (all NaN) — every row not finished
    int bp = 0x7fffffff, bs = 0;
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int ps = s_pos[lane + 64 * s];
      const bool c = (ps >= g) & (wm >= 0.0 ? __builtin_fabs(col[s]) == wm : true);
      const bool take = c & (ps < bp);
      bp = take ? ps : bp;
      bs = take ? s : bs;
    }
    int q = bp;
    double dummy = 0.0;
    tl_argmax_stage<kDppQuadXor1>(dummy, q);
    tl_argmax_stage<kDppQuadXor2>(dummy, q);
    tl_argmax_stage<kDppRowHalfMirror>(dummy, q);
    tl_argmax_stage<kDppRowMirror>(dummy, q);
    int bq = __builtin_amdgcn_readlane(q, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) { const int oq = __builtin_amdgcn_readlane(q, 16 * r); bq = oq < bq ? oq : bq; }
    ls = __ffsll((long long)__ballot(bp == bq)) - 1;
    ss = __builtin_amdgcn_readlane(bs, ls);
    piv = tl_readlane_f64(tl_slot<RS>(col, ss), ls);
  }
  const bool zero = piv == 0.0;
  const double inv = zero ? 0.0 : div_refined_rcp(piv);  // a zero pivot eliminates nothing (its column is all zeros)
#pragma unroll
  for (int s = 0; s < RS; ++s) lcol[64 * s] = col[s] * inv;
  if (lane == ls) lcol[64 * ss] = 1.0;  // the pivot row itself: its entries behind this column become exact zeros
  const int rstar = ls + 64 * ss;
  const int ps = s_pos[rstar];
  if (lane == 0) {
    s_hdr[2 * (k & 3)] = ls;
    s_hdr[2 * (k & 3) + 1] = ss;
    dyn[tl_ubuf<RS>() + k * (tl_cfg<RS>::kSW + 1) + k] = piv;
    if (zero) s_flags[0] = 1;
    s_pos[rstar] = (short)g;  // the row at the diagonal position trades places with the winner
    if (rg != rstar) { s_pos[rg] = (short)ps; s_rowat[ps] = (short)rg; }
    s_rowat[g] = (short)rstar;
    s_prow[pbase + k] = rstar;
    s_ipiv[pbase + k] = ps;
  }
#ifdef TL_X_STEPPROF
  if (threadIdx.x == 0 && blockIdx.x == 0) { tl_stepprof[4] += __builtin_readcyclecounter() - ts0_; tl_stepprof[5] += 1; }
#endif
}

// c -= u l (the elimination of one step in one register column)
template <int RS>
__device__ __forceinline__ void tl_co_elim(tl_col<RS>& c, const double (&l)[RS], double u) {
#pragma unroll
  for (int s = 0; s < RS; ++s) c[s] = __builtin_fma(-u, l[s], c[s]);
}
// Pivot step k = NW JO + wo of the sub-panel at column cb, behind the barrier that published it (NW wavefronts; JO static: register column of its
// owner; wavefront w holds the sub-panel's columns w, w + NW, ...): every wavefront eliminates in its columns behind k.  The chain that bounds the panel
// is barrier -> multipliers -> column k + 1 -> search of step k + 1 -> publish -> barrier, so the wavefront that owns column k + 1 (pipe == true) does
// exactly that and PUTS OFF the elimination of step k in its other columns: it makes up for it behind the next barrier, before step k + 1's (the same
// multiply-adds in the same order, so every entry sees the operations it would see without the delay).  What it needs then is still there: the
// multipliers in Lbuf (a column is reused 16 steps later), {lane, slot} of step k's pivot row in s_hdr (four entries deep), and the pivot row's
// entries in its own registers (a finished row changes only through its own step's elimination).
template <int RS, int NW, int JO>
__device__ __forceinline__ void tl_co_step(tl_col<RS> (&a)[tl_cfg<RS>::kSW / NW], int wo, int ws, bool pipe, int cb, int pbase, double* __restrict__ dyn, short* s_pos, short* s_rowat,
                                           int* s_prow, int* s_ipiv, int* s_hdr, int* s_flags, int wave, int lane) {
  constexpr int P = 64 * RS, CPW = tl_cfg<RS>::kSW / NW, UP = tl_cfg<RS>::kSW + 1;
  const int k = NW * JO + wo;
  const int kk = k & (kTlLC - 1);
  const double* const lcol = dyn + kk * P + lane;
  double* const ubuf = dyn + tl_ubuf<RS>() + wave;
#ifdef TL_X_STEPPROF
  unsigned long long t0_ = __builtin_readcyclecounter();
#define TL_T(ix) { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && blockIdx.x == 0) tl_stepprof[ix] += now_ - t0_; t0_ = now_; }
#else
#define TL_T(ix)
#endif
  const int ls = __builtin_amdgcn_readfirstlane(s_hdr[2 * (k & 3)]), ss = __builtin_amdgcn_readfirstlane(s_hdr[2 * (k & 3) + 1]);
  double l[RS];
#pragma unroll
  for (int s = 0; s < RS; ++s) l[s] = lcol[64 * s];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  TL_T(0)
  // this wavefront owned step k and ran its search at once (see above): step k - 1 is still to be eliminated in its columns behind k
  if constexpr (JO < CPW - 1) {
    if (wave == wo && kk != 0) {
      const int k1 = k - 1;
      const int ls1 = __builtin_amdgcn_readfirstlane(s_hdr[2 * (k1 & 3)]), ss1 = __builtin_amdgcn_readfirstlane(s_hdr[2 * (k1 & 3) + 1]);
      const double* const l1col = dyn + (k1 & (kTlLC - 1)) * P + lane;
      double l1[RS];
#pragma unroll
      for (int s = 0; s < RS; ++s) l1[s] = l1col[64 * s];
#pragma unroll
      for (int j = JO + 1; j < CPW; ++j) {
        const double u1 = tl_readlane_f64(tl_slot<RS>(a[j], ss1), ls1);
        if (lane == 0) ubuf[k1 * UP + NW * j] = u1;  // U11 of step k - 1 in this column
        tl_co_elim<RS>(a[j], l1, u1);
      }
    }
  }
  TL_T(1)
  const bool next_owner = pipe && wave == (wo + 1) % NW && k + 1 < ws;
  if (next_owner) {  // the pivot row's entry in column k + 1 only (the row is finished: out of this wavefront's own registers), that column, the search
    if (wo == NW - 1) {  // wavefront 0, its next register column
      if constexpr (JO < CPW - 1) {
        const double u = tl_readlane_f64(tl_slot<RS>(a[JO + 1], ss), ls);
        if (lane == 0) ubuf[k * UP + NW * (JO + 1)] = u;
        tl_co_elim<RS>(a[JO + 1], l, u);
        tl_co_search<RS>(a[JO + 1], k + 1, cb, pbase, dyn, s_pos, s_rowat, s_prow, s_ipiv, s_hdr, s_flags, lane);
      }
    } else {
      const double u = tl_readlane_f64(tl_slot<RS>(a[JO], ss), ls);
      if (lane == 0) ubuf[k * UP + NW * JO] = u;
      tl_co_elim<RS>(a[JO], l, u);
      tl_co_search<RS>(a[JO], k + 1, cb, pbase, dyn, s_pos, s_rowat, s_prow, s_ipiv, s_hdr, s_flags, lane);
    }
  } else {
    // the pivot row's entries in this wavefront's columns behind k (U11), and the elimination there
    if (wave > wo) {
      const double u = tl_readlane_f64(tl_slot<RS>(a[JO], ss), ls);
      if (lane == 0) ubuf[k * UP + NW * JO] = u;
      tl_co_elim<RS>(a[JO], l, u);
    }
#pragma unroll
    for (int j = JO + 1; j < CPW; ++j) {
      const double u = tl_readlane_f64(tl_slot<RS>(a[j], ss), ls);
      if (lane == 0) ubuf[k * UP + NW * j] = u;
      tl_co_elim<RS>(a[j], l, u);
    }
  }
  TL_T(2)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: never wait for global stores here
  TL_T(3)
}

// Multipliers (and, for the rows chosen in this sub-panel, their U entries from their step on) of the steps [k0, k0 + cnt) from LDS to the rows of W.
template <int RS, int NW>
__device__ __forceinline__ void tl_flush(tl_gdouble* __restrict__ W, int ldw, int n, int cb, int k0, int cnt, const double* __restrict__ dyn, const short* s_pos, int tid) {
  constexpr int P = 64 * RS, SW = tl_cfg<RS>::kSW;
#pragma unroll
  for (int i = 0; i < RS / NW; ++i) {
    const int row = tid + 64 * NW * i;
    const int pr = row < n ? (int)s_pos[row] : -1;
    if (pr >= cb) {  // the row entered this sub-panel
      const int kr = pr - cb;  // its own step, if it was chosen here (else >= SW or beyond the steps done)
      tl_gd2* dst = reinterpret_cast<tl_gd2*>(W + (size_t)row * ldw + cb + k0);
      for (int c = 0; c < cnt; c += 2) {
        tl_d2 v;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int k = k0 + c + e;
          v[e] = k < kr ? dyn[(k & (kTlLC - 1)) * P + row] : dyn[tl_ubuf<RS>() + (kr < SW ? kr : 0) * (SW + 1) + k];
        }
        dst[c >> 1] = v;
      }
    }
  }
}

// Inverse of the unit lower triangular 16 x 16 diagonal block of the 16 steps whose multipliers Lbuf holds (panel steps pb .. pb + 15, diagonal block `blk` of the
// panel), by the first 16 lanes of one wavefront: lane = column.  The inverses are the A operands of every blocked substitution that follows (the later
// sub-panels' U', the trailing phase's U12).
template <int RS>
__device__ __forceinline__ void tl_invert_diag(double* __restrict__ dyn, const int* s_prow, int pb, int blk, int lane) {
  constexpr int P = 64 * RS;
  if (lane < 16) {
    int rows[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rows[r] = s_prow[pb + r];
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = r == lane ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 15; ++i)
#pragma unroll
      for (int r = i + 1; r < 16; ++r) x[r] = __builtin_fma(-dyn[i * P + rows[r]], x[i], x[r]);
    double* const invd = dyn + tl_invd<RS>() + blk * 272;
#pragma unroll
    for (int r = 0; r < 16; ++r) invd[r * 17 + lane] = x[r];
  }
}

// The panel of 64 columns at jb (sub-panels of kSW columns).  Not inlined: its registers are allocated apart from the rest of the kernel.
template <int RS>
__device__ __noinline__ void tl_panel(double* __restrict__ W_generic_, int ldw_, int n_, int jb_, double* dyn_, short* s_pos_, short* s_rowat_, int* s_prow_, int* s_ipiv_,
                                      int* s_hdr_, int* s_flags_, const unsigned short* s_rowlist_, int m_in_, unsigned long long* phase_clocks_) {
  double* const W_generic = tl_uni(W_generic_);
  const int ldw = tl_uni(ldw_), n = tl_uni(n_), jb = tl_uni(jb_), m_in = tl_uni(m_in_);
  double* const dyn = tl_uni(dyn_);
  short* const s_pos = tl_uni(s_pos_);
  short* const s_rowat = tl_uni(s_rowat_);
  int* const s_prow = tl_uni(s_prow_);
  int* const s_ipiv = tl_uni(s_ipiv_);
  int* const s_hdr = tl_uni(s_hdr_);
  int* const s_flags = tl_uni(s_flags_);
  const unsigned short* const s_rowlist = tl_uni(s_rowlist_);
  unsigned long long* const phase_clocks = tl_uni(phase_clocks_);
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  auto mark = [&](int phase) {
    if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[phase] += now - tprev; tprev = now; }
  };
  constexpr int NW = tl_cfg<RS>::kWaves, NT = 64 * NW, SW = tl_cfg<RS>::kSW, CPW = SW / NW;
  constexpr int R = RS / NW;       // rows per thread in the staging layout (thread t: rows t, t + NT, ...)
  constexpr int P = 64 * RS;       // rows of a column in LDS
  constexpr int NH = SW / kTlLC;   // the sub-panel's columns pass through LDS 16 at a time
  constexpr int HC = kTlLC;
  constexpr int NCT = SW / 16;     // column tiles of a sub-panel
  constexpr int MAXD = (kTlPW - SW) / 16;  // diagonal blocks in front of the last sub-panel
  tl_gdouble* const W = (tl_gdouble*)W_generic;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  [[maybe_unused]] double* const Us = dyn + kTlUs;  // [<= 48][kTlUsP]: dead before the stage writes T over it (unused by the fused stage)
  const double* const invd = dyn + tl_invd<RS>();
#pragma nounroll
  for (int sp = 0; sp < kTlPW / SW; ++sp) {
    const int cb = jb + SW * sp;
    const int ws = (n - cb) < SW ? (n - cb) : SW;
    if (ws <= 0) break;
#if DSH_TL_FUSED_STAGE
    tl_col<RS> a[CPW];
    // element (column c of the 16 in LDS, row) of the transposing stage T: the row index is permuted inside its aligned block of 32 by the column, so that the
    // matrix-core layout (16 lanes = 16 columns of one row) and the two thread-per-row / column-per-wavefront layouts are all free of bank conflicts
    auto tsw = [&](int c, int row) { return c * P + (row ^ (c << 1)); };
    if (sp > 0) {
      // ---- the columns cb..cb+SW-1 take the D = SW sp eliminations of the panel's earlier sub-panels on the matrix cores, ON THE WAY into the stage: U' = L11A^-1 B
      // for the pivot rows (blocked substitution; EVERY wavefront solves it, so that U' is where the update wants its B operand: in the accumulators), then
      // B - L_A U' of the rows still unfinished goes straight into T — no round trip of the sub-panel through W (rows finished since the panel began are not
      // staged; the flush writes these columns of every staged row anyway).
      const int D = SW * sp, nd = D / 16;
      double* const la = dyn;         // [D][kTlLaP]
      const int q = lane >> 4, j = lane & 15;
      // Inside every block of 16 eliminations the matrix-core index 4 g + q (k-block g, lane quarter q) stands for elimination 4 q + g: a lane then needs FOUR ADJACENT
      // multipliers of a row per block (two 16-byte loads) instead of four at a stride of four (four 8-byte loads that touch the same lines four times over).  U' is solved in
      // the same labelling (rows of its blocks, rows and columns of L11A and of the inverses), so operands and accumulators agree.
      const int pj = 4 * (j & 3) + (j >> 2);
      for (int idx = tid; idx < D * D; idx += NT) {
        const int k = idx / D, i = idx - k * D;
        if (i < (k & ~15)) la[k * kTlLaP + i] = W[(size_t)s_prow[k] * ldw + jb + i];
      }
      __syncthreads();
      tl_d4 X[NCT][MAXD];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        const int c0 = 16 * ct;
        tl_d4 B[MAXD];
#pragma unroll
        for (int rb = 0; rb < MAXD; ++rb)
          if (rb < nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) B[rb][r] = W[(size_t)s_prow[16 * rb + 4 * q + r] * ldw + cb + c0 + j];
          }
#pragma unroll
        for (int rb = 0; rb < MAXD; ++rb) {
          if (rb < nd) {
#pragma unroll
            for (int cbk = 0; cbk < rb; ++cbk)
#pragma unroll
              for (int kb = 0; kb < 4; ++kb) B[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[(16 * rb + pj) * kTlLaP + 16 * cbk + 4 * q + kb], X[ct][cbk][kb], B[rb], 0, 0, 0);
            tl_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(invd[rb * 272 + pj * 17 + 4 * q + kb], B[rb][kb], acc, 0, 0, 0);
            X[ct][rb] = acc;
          } else {
            X[ct][rb] = tl_d4{0.0, 0.0, 0.0, 0.0};
          }
        }
      }
      __syncthreads();  // L11A is dead: T lies over it; every wavefront has read the pivot rows' entries, one writes U' over them
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        if (wave == ct % NW) {
#pragma unroll
          for (int rb = 0; rb < MAXD; ++rb)
            if (rb < nd) {
#pragma unroll
              for (int r = 0; r < 4; ++r) W[(size_t)s_prow[16 * rb + 4 * q + r] * ldw + cb + 16 * ct + j] = X[ct][rb][r];
            }
        }
      mark(4);
      const int nrt = (m_in + 15) / 16;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i = 0; i < R; ++i) {  // finished rows and rows beyond n enter as zeros
          const int row = tid + NT * i;
          if (!(row < n && s_pos[row] >= cb)) {
#pragma unroll
            for (int c = 0; c < HC; ++c) dyn[tsw(c, row)] = 0.0;
          }
        }
        // the row tiles dealt to the wavefronts, TU of them at a time: their loads are all in flight before the first is multiplied
        constexpr int TU = DSH_TL_STAGE_TU;
        for (int tile0 = wave; tile0 < nrt; tile0 += NW * TU) {
          double aneg[TU][4 * MAXD];
          tl_d4 acc[TU];
          int rows[TU][4];
          unsigned ok[TU];
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const int tile = tile0 + NW * u;
            ok[u] = 0;
            if (tile < nrt) {
              const size_t arow = (size_t)s_rowlist[16 * tile + j] * ldw;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                rows[u][r] = s_rowlist[16 * tile + 4 * r + q];
                acc[u][r] = W[(size_t)rows[u][r] * ldw + cb + 16 * h + j];
                if (16 * tile + 4 * r + q < m_in && s_pos[rows[u][r]] >= cb) ok[u] |= 1u << r;
              }
#pragma unroll
              for (int rb = 0; rb < MAXD; ++rb)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  tl_d2 v = {0.0, 0.0};
                  if (rb < nd) v = *reinterpret_cast<const tl_gd2*>(W + arow + jb + 16 * rb + 4 * q + 2 * e);
                  aneg[u][4 * rb + 2 * e] = -v[0];
                  aneg[u][4 * rb + 2 * e + 1] = -v[1];
                }
            }
          }
#pragma unroll
          for (int kb = 0; kb < 4 * MAXD; ++kb)
            if (kb < 4 * nd) {
#pragma unroll
              for (int u = 0; u < TU; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(aneg[u][kb], X[h][kb >> 2][kb & 3], acc[u], 0, 0, 0);
            }
#pragma unroll
          for (int u = 0; u < TU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (ok[u] & (1u << r)) dyn[tsw(j, rows[u][r])] = acc[u][r];
        }
        __syncthreads();
        mark(9);
#pragma unroll
        for (int jj = 0; jj < CPW / NH; ++jj)
#pragma unroll
          for (int s = 0; s < RS; ++s) a[(CPW / NH) * h + jj][s] = dyn[tsw(wave + NW * jj, lane + 64 * s)];
        __syncthreads();
        mark(10);
      }
    } else {
      mark(4);
      // ---- stage of the panel's first sub-panel: thread per row in, column per wavefront out
#pragma unroll
      for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i = 0; i < R; ++i) {  // HC columns of one row at a time (registers); finished rows and rows beyond n enter as zeros
          const int row = tid + NT * i;
          const bool live = row < n && s_pos[row] >= cb;
          double b[HC];
          if (live) {
            const tl_gd2* src = reinterpret_cast<const tl_gd2*>(W + (size_t)row * ldw + cb + HC * h);
#pragma unroll
            for (int c = 0; c < HC; c += 2) { const tl_d2 v = src[c >> 1]; b[c] = v[0]; b[c + 1] = v[1]; }
          }
#pragma unroll
          for (int c = 0; c < HC; ++c) dyn[tsw(c, row)] = live ? b[c] : 0.0;
        }
        __syncthreads();
        mark(9);
#pragma unroll
        for (int jj = 0; jj < CPW / NH; ++jj)
#pragma unroll
          for (int s = 0; s < RS; ++s) a[(CPW / NH) * h + jj][s] = dyn[tsw(wave + NW * jj, lane + 64 * s)];
        __syncthreads();
        mark(10);
      }
    }
#else
    if (sp > 0) {
      // ---- the columns cb..cb+SW-1 take the D = SW sp eliminations of the panel's earlier sub-panels, on the matrix cores (LDS broadcast reads made the
      // vector form of this 50 us per panel): the blocks of L11A below its diagonal blocks to LDS (the inverses of the diagonal blocks are there since
      // their sub-panels finished), U' = L11A^-1 B for the pivot rows (blocked substitution, one column tile per wavefront), then B -= L_A U' for the
      // rows that entered the panel (the row list of the previous panel; rows finished since are not stored).
      const int D = SW * sp, nd = D / 16;
      double* const la = dyn;         // [D][kTlLaP]
      const int q = lane >> 4, j = lane & 15;
      for (int idx = tid; idx < D * D; idx += NT) {
        const int k = idx / D, i = idx - k * D;
        if (i < (k & ~15)) la[k * kTlLaP + i] = W[(size_t)s_prow[k] * ldw + jb + i];
      }
      __syncthreads();
      if (wave < NCT) {
        const int c0 = 16 * wave;
        tl_d4 B[MAXD], X[MAXD];
#pragma unroll
        for (int rb = 0; rb < MAXD; ++rb)
          if (rb < nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) B[rb][r] = W[(size_t)s_prow[16 * rb + 4 * r + q] * ldw + cb + c0 + j];
          }
#pragma unroll
        for (int rb = 0; rb < MAXD; ++rb)
          if (rb < nd) {
#pragma unroll
            for (int cbk = 0; cbk < rb; ++cbk)
#pragma unroll
              for (int kb = 0; kb < 4; ++kb) B[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-la[(16 * rb + j) * kTlLaP + 16 * cbk + 4 * kb + q], X[cbk][kb], B[rb], 0, 0, 0);
            tl_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(invd[rb * 272 + j * 17 + 4 * kb + q], B[rb][kb], acc, 0, 0, 0);
            X[rb] = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Us[(16 * rb + 4 * r + q) * kTlUsP + c0 + j] = acc[r];
              W[(size_t)s_prow[16 * rb + 4 * r + q] * ldw + cb + c0 + j] = acc[r];
            }
          }
      }
      __syncthreads();
      const int nrt = (m_in + 15) / 16;
      for (int tile = wave; tile < nrt; tile += NW) {
        double aneg[4 * MAXD];
        const size_t arow = (size_t)s_rowlist[16 * tile + j] * ldw;
#pragma unroll
        for (int kb = 0; kb < 4 * MAXD; ++kb) aneg[kb] = kb < 4 * nd ? -W[arow + jb + 4 * kb + q] : 0.0;
        size_t ro[4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = s_rowlist[16 * tile + 4 * r + q];
          ro[r] = (size_t)row * ldw + cb + j;
          ok[r] = 16 * tile + 4 * r + q < m_in && s_pos[row] >= cb;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          tl_d4 acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = W[ro[r] + 16 * ct];
#pragma unroll
          for (int kb = 0; kb < 4 * MAXD; ++kb)
            if (kb < 4 * nd) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aneg[kb], Us[(4 * kb + q) * kTlUsP + 16 * ct + j], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) if (ok[r]) W[ro[r] + 16 * ct] = acc[r];
        }
      }
      __syncthreads();
    }
    mark(4);
    // ---- stage: thread per row in (the second sub-panel's rows take the 32 eliminations of the first on the way: row -= L_row U'), column per wavefront out
    tl_col<RS> a[CPW];
    {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int i = 0; i < R; ++i) {  // HC columns of one row at a time (registers); finished rows and rows beyond n enter as zeros
          const int row = tid + NT * i;
          const bool live = row < n && s_pos[row] >= cb;
          double b[HC];
          if (live) {
            const tl_gd2* src = reinterpret_cast<const tl_gd2*>(W + (size_t)row * ldw + cb + HC * h);
#pragma unroll
            for (int c = 0; c < HC; c += 2) { const tl_d2 v = src[c >> 1]; b[c] = v[0]; b[c + 1] = v[1]; }
          }
#pragma unroll
          for (int c = 0; c < HC; ++c) dyn[c * P + row] = live ? b[c] : 0.0;
        }
        __syncthreads();
        mark(9);
#pragma unroll
        for (int jj = 0; jj < CPW / NH; ++jj)
#pragma unroll
          for (int s = 0; s < RS; ++s) a[(CPW / NH) * h + jj][s] = dyn[(wave + NW * jj) * P + lane + 64 * s];
        __syncthreads();
        mark(10);
      }
    }
#endif
    mark(5);
    // ---- the pivot steps
    const int pbase = SW * sp;
#define TL_STEPS(JO)                                                                                                                        \
  if constexpr (JO < CPW) {                                                                                                                 \
    _Pragma("nounroll") for (int wo = 0; wo < NW; ++wo) {                                                                                   \
      if (NW * JO + wo >= ws) break;                                                                                                        \
      const bool pipe = NW * JO + wo != kTlLC - 1; /* the next column of Lbuf is free only behind the flush */                              \
      tl_co_step<RS, NW, JO>(a, wo, ws, pipe, cb, pbase, dyn, s_pos, s_rowat, s_prow, s_ipiv, s_hdr, s_flags, wave, lane);                  \
    }                                                                                                                                       \
    if (SW > kTlLC && JO == kTlLC / NW - 1 && ws > kTlLC) {                                                                                 \
      __syncthreads();                                                                                                                      \
      tl_flush<RS, NW>(W, ldw, n, cb, 0, kTlLC, dyn, s_pos, tid);                                                                           \
      if (wave == 1) tl_invert_diag<RS>(dyn, s_prow, pbase, pbase / 16, lane);                                                              \
      __syncthreads();                                                                                                                      \
      if (wave == 0) tl_co_search<RS>(a[kTlLC / NW < CPW ? kTlLC / NW : 0], kTlLC, cb, pbase, dyn, s_pos, s_rowat, s_prow, s_ipiv, s_hdr, s_flags, lane); \
      __syncthreads();                                                                                                                      \
    }                                                                                                                                       \
  }
    if (wave == 0) tl_co_search<RS>(a[0], 0, cb, pbase, dyn, s_pos, s_rowat, s_prow, s_ipiv, s_hdr, s_flags, lane);
    __syncthreads();
    TL_STEPS(0) TL_STEPS(1) TL_STEPS(2) TL_STEPS(3) TL_STEPS(4) TL_STEPS(5) TL_STEPS(6) TL_STEPS(7)
#undef TL_STEPS
    __syncthreads();
    mark(6);
    const int f0 = ws <= kTlLC ? 0 : kTlLC;  // the steps whose multipliers Lbuf holds now
    tl_flush<RS, NW>(W, ldw, n, cb, f0, ((ws + 1) & ~1) - f0, dyn, s_pos, tid);
    if (wave == 1 && ws - f0 == 16) tl_invert_diag<RS>(dyn, s_prow, pbase + f0, (pbase + f0) / 16, lane);
    __syncthreads();
    mark(7);
  }
}

// Everything behind a finished 64-column panel: U12 and the update of the active rows, a chunk of <= kCH trailing columns at a time.  A function of its own
// (not inlined) so that its registers — the L21 operand of kRT row tiles stays in them across the chunks — are allocated apart from the panel's.
template <int RS>
__device__ __noinline__ void tl_trailing(double* __restrict__ W_generic_, double* __restrict__ F_generic_, double* dyn_, const int* s_prow_, const unsigned short* s_rowlist_,
                                         unsigned long long* phase_clocks_, int n_, int ldw_, int jb_, int nct_, int m2_) {
  using C = tl_cfg<RS>;
  double* const W_generic = tl_uni(W_generic_);
  double* const F_generic = tl_uni(F_generic_);
  double* const dyn = tl_uni(dyn_);
  const int* const s_prow = tl_uni(s_prow_);
  const unsigned short* const s_rowlist = tl_uni(s_rowlist_);
  unsigned long long* const phase_clocks = tl_uni(phase_clocks_);
  const int n = tl_uni(n_), ldw = tl_uni(ldw_), jb = tl_uni(jb_), nct = tl_uni(nct_), m2 = tl_uni(m2_);
  constexpr int LDP = C::kLDP, CH = C::kCH, RTM = C::kRT, NW = C::kWaves;
  constexpr int PF = C::kPF;
  tl_gdouble* const W = (tl_gdouble*)W_generic;
  tl_gdouble* const F = (tl_gdouble*)F_generic;
  double* const u12s = dyn;
  const double* const l11 = dyn + 64 * LDP;        // rows 16..63, columns 0..47 of L11, pitch kTlL11P
  const double* const invd = dyn + tl_invd<RS>();  // the inverses of the panel's four diagonal blocks (tl_invert_diag)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  auto mark = [&](int phase) {
    if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[phase] += now - tprev; tprev = now; }
  };
  const int q = lane >> 4, j = lane & 15;
  // decomposition of the update: groups of rtw row tiles; with fewer groups than wavefronts the column tiles are split as well
  const int nrt = (m2 + 15) / 16;
  const int rtw = (nrt + NW - 1) / NW < RTM ? (nrt + NW - 1) / NW : RTM;
  const int groups = (nrt + rtw - 1) / rtw;
  const int csplit = groups >= NW ? 1 : NW / groups;
  const bool fixed_group = groups <= NW;
  double aneg[RTM][16];
  unsigned roffb[RTM][4];
  unsigned valid = 0;
  int loaded_group = -1;
  for (int c_lo = jb + kTlPW; c_lo < nct; c_lo += CH) {
    const int cw = (nct - c_lo) < CH ? (nct - c_lo) : CH;
    const int ntc = cw / 16;
    // ---- U12 of the chunk: blocked substitution on the matrix cores, one column tile per wavefront at a time
    {
      // the A operands (inverses of the diagonal blocks, negated blocks below them) come from LDS as they are needed: holding all 40 of them next
      // to the L21 operand of the update, which stays in registers across the chunks, does not fit in the registers
      const double* const dinv_l = invd + j * 17 + q;            // block b, k-block kb: + b * 272 + 4 kb
      const double* const l11_l = l11 + j * kTlL11P + q;         // block (rb, cbk), rb >= 1, k-block kb: + 16 (rb - 1) * pitch + 16 cbk + 4 kb
      for (int tc = wave; tc < ntc; tc += NW) {
        const int c0 = c_lo + 16 * tc;
        tl_d4 B[4], X[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) B[rb][r] = W[(size_t)(s_prow[16 * rb + 4 * r + q] * ldw) + c0 + j];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
          for (int cbk = 0; cbk < rb; ++cbk) {
            double lo[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) lo[kb] = -l11_l[16 * (rb - 1) * kTlL11P + 16 * cbk + 4 * kb];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) B[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo[kb], X[cbk][kb], B[rb], 0, 0, 0);
          }
          double di[4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) di[kb] = dinv_l[rb * 272 + 4 * kb];
          tl_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(di[kb], B[rb][kb], acc, 0, 0, 0);
          X[rb] = acc;
        }
        const bool incol = c0 + j < n;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            u12s[(16 * rb + 4 * r + q) * LDP + 16 * tc + j] = X[rb][r];
            if (incol) F[(size_t)(c0 + j) * n + jb + 16 * rb + 4 * r + q] = X[rb][r];
          }
      }
    }
    __syncthreads();
    mark(2);
    // ---- A22 -= L21 U12 for the chunk's columns
    for (int grp = fixed_group ? wave % groups : wave; grp < groups; grp += NW) {
      const int csub = fixed_group ? wave / groups : 0;
      if (csub >= csplit) break;
      if (grp != loaded_group) {
        loaded_group = grp;
        valid = 0;
#pragma unroll
        for (int t = 0; t < RTM; ++t) {
          const int tile = grp * rtw + t;
          const bool tv = t < rtw && tile < nrt;
          const int arow = tv ? (int)s_rowlist[16 * tile + j] * ldw : 0;
#pragma unroll
          for (int kb = 0; kb < 16; ++kb) aneg[t][kb] = tv ? -W[(size_t)arow + jb + 4 * kb + q] : 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            roffb[t][r] = tv ? (unsigned)s_rowlist[16 * tile + 4 * r + q] * (unsigned)ldw * 8u : 0u;
            if (tv && 16 * tile + 4 * r + q < m2) valid |= 1u << (4 * t + r);
          }
        }
      }
      const int ngt = (nrt - grp * rtw) < rtw ? (nrt - grp * rtw) : rtw;  // row tiles of this group
      if constexpr (RTM == 3) {
        switch (ngt) {
          case 1: tl_update_tiles<1, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          case 2: tl_update_tiles<2, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          default: tl_update_tiles<3, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
        }
      } else if constexpr (RTM == 4) {
        switch (ngt) {
          case 1: tl_update_tiles<1, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          case 2: tl_update_tiles<2, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          case 3: tl_update_tiles<3, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          default: tl_update_tiles<4, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
        }
      } else {
        if (ngt == 1) tl_update_tiles<1, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane);
        else tl_update_tiles<2, RTM, LDP, PF>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane);
      }
      if (fixed_group) break;
    }
    __syncthreads();
    mark(3);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The trailing phase with the COLUMNS dealt to the wavefronts (round 5).  A wavefront takes CPW <= kCT adjacent column tiles: it computes their U12 by the
// blocked substitution — the accumulators it ends up with ARE the B operands of the update (see dsh_lu_tiled.hpp) — keeps them in registers (16 doubles per
// column tile), writes them to F, and then walks down the row tiles of the active rows: L21 operand of the row tile (16 doubles, straight from W: every
// wavefront of the workgroup walks the same row tiles, so all but the first read hits the CU's cache), the C tiles, 16 matrix-core instructions per tile,
// store.  No U12 chunk in LDS, no barrier inside the phase (the wavefronts' tiles are disjoint, what they read — pivot rows, the panel's columns — nobody
// writes), and a wavefront's C accesses are CPW x 128 adjacent bytes of a row.  With fewer column groups than wavefronts the row tiles of a group are dealt to
// several wavefronts (each repeats the group's U12: 40 instructions per tile; one writes F).
// Barrier for data that went through LDS only: __syncthreads() also waits for every global load and STORE in flight (vmcnt(0)) — in the trailing phase that drained
// the C tiles' prefetch and made every chunk wait for the write latency of the tiles just stored.
__device__ __forceinline__ void tl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int RS, int CPW>
__device__ __forceinline__ void tl_colgroup(tl_gdouble* __restrict__ W, tl_gdouble* __restrict__ F, double* __restrict__ l21s, const double* __restrict__ l11,
                                            const double* __restrict__ invd, const int* s_prow, const unsigned short* s_rowlist, int n, int ldw, int jb, int c_first, int rs,
                                            int rsplit, int nrt, int m2, int wave, int lane, bool active) {
  const int q = lane >> 4, j = lane & 15;
  tl_gchar* const Wb = reinterpret_cast<tl_gchar*>(W);
  tl_d4 X[CPW][4];
  if (active) {
    const double* const dinv_l = invd + j * 17 + q;
    const double* const l11_l = l11 + j * kTlL11P + q;
    unsigned prow_off[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) prow_off[e] = (unsigned)s_prow[4 * e + q] * (unsigned)ldw * 8u;  // rows 16 rb + 4 r + q, e = 4 rb + r
#pragma unroll
    for (int ct = 0; ct < CPW; ++ct) {
      const int c0 = c_first + 16 * ct;
      tl_d4 B[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) B[rb][r] = *reinterpret_cast<const tl_gdouble*>(Wb + (prow_off[4 * rb + r] + (unsigned)(c0 + j) * 8u));
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int cbk = 0; cbk < rb; ++cbk) {
          double lo[4];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) lo[kb] = -l11_l[16 * (rb - 1) * kTlL11P + 16 * cbk + 4 * kb];
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) B[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo[kb], X[ct][cbk][kb], B[rb], 0, 0, 0);
        }
        double di[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) di[kb] = dinv_l[rb * 272 + 4 * kb];
        tl_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(di[kb], B[rb][kb], acc, 0, 0, 0);
        X[ct][rb] = acc;
      }
      if (rs == 0 && c0 + j < n) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) F[(size_t)(c0 + j) * n + jb + 16 * rb + 4 * r + q] = X[ct][rb][r];
      }
    }
  }
  // ---- the update.  The L21 operand goes through LDS, 64 rows at a time: a row is read ONCE per workgroup as 512 adjacent bytes (wavefront w: rows w, w + NW, ...
  // of the chunk; the next chunk's rows are in flight while this one is multiplied) and every wavefront takes its 16 x 4 operand pieces out of LDS.  Read per
  // wavefront straight from W the same operand is 16 loads of 16 lines each per row tile — measured (DSH_TL_X_NOA): a fifth of the whole factorisation's time went
  // into those line look-ups.  The C tiles of the next row tile are in flight while one is multiplied.
  constexpr int NW = tl_cfg<RS>::kWaves, SR = 64 / NW, LP = kTlL21P;
  const unsigned colb = (unsigned)(c_first + j) * 8u;
  const int nchunk = (nrt + 3) / 4;
  double st[SR];
  auto stage_load = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < SR; ++i) {
      int idx = 64 * chunk + wave + NW * i;
      idx = idx < m2 ? idx : m2 - 1;
      st[i] = -W[(size_t)s_rowlist[idx] * ldw + jb + lane];
    }
  };
  tl_d4 c_cur[CPW], c_nxt[CPW];
  unsigned ro_cur[4], ro_nxt[4];
  unsigned v_cur = 0, v_nxt = 0;
  auto fetch = [&](int tile) {
    v_nxt = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ro_nxt[r] = (unsigned)s_rowlist[16 * tile + 4 * r + q] * (unsigned)ldw * 8u + colb;
      if (16 * tile + 4 * r + q < m2) v_nxt |= 1u << r;
    }
#pragma unroll
    for (int ct = 0; ct < CPW; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) c_nxt[ct][r] = *reinterpret_cast<const tl_gdouble*>(Wb + (ro_nxt[r] + 128u * (unsigned)ct));
  };
  stage_load(0);
  int tile = rs;
  if (active && tile < nrt) fetch(tile);
  for (int chunk = 0; chunk < nchunk; ++chunk) {
#pragma unroll
    for (int i = 0; i < SR; ++i) l21s[(wave + NW * i) * LP + lane] = st[i];
    tl_lds_barrier();
    if (chunk + 1 < nchunk) stage_load(chunk + 1);
    if (active) {
      const int tend = 4 * chunk + 4 < nrt ? 4 * chunk + 4 : nrt;
      for (; tile < tend; tile += rsplit) {
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct) c_cur[ct] = c_nxt[ct];
#pragma unroll
        for (int r = 0; r < 4; ++r) ro_cur[r] = ro_nxt[r];
        v_cur = v_nxt;
        if (tile + rsplit < nrt) fetch(tile + rsplit);
        const double* const ap = l21s + (16 * (tile & 3) + j) * LP + q;
        double a[16];
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) a[kb] = ap[4 * kb];
#pragma unroll
        for (int kb = 0; kb < 16; ++kb)
#pragma unroll
          for (int ct = 0; ct < CPW; ++ct) c_cur[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb], X[ct][kb >> 2][kb & 3], c_cur[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (v_cur & (1u << r)) *reinterpret_cast<tl_gdouble*>(Wb + (ro_cur[r] + 128u * (unsigned)ct)) = c_cur[ct][r];
      }
    }
    tl_lds_barrier();
  }
}

template <int RS>
__device__ __noinline__ void tl_trailing_cols(double* __restrict__ W_generic_, double* __restrict__ F_generic_, double* dyn_, const int* s_prow_, const unsigned short* s_rowlist_,
                                              unsigned long long* phase_clocks_, int n_, int ldw_, int jb_, int nct_, int m2_) {
  using C = tl_cfg<RS>;
  double* const W_generic = tl_uni(W_generic_);
  double* const F_generic = tl_uni(F_generic_);
  double* const dyn = tl_uni(dyn_);
  const int* const s_prow = tl_uni(s_prow_);
  const unsigned short* const s_rowlist = tl_uni(s_rowlist_);
  unsigned long long* const phase_clocks = tl_uni(phase_clocks_);
  const int n = tl_uni(n_), ldw = tl_uni(ldw_), jb = tl_uni(jb_), nct = tl_uni(nct_), m2 = tl_uni(m2_);
  constexpr int NW = C::kWaves, CT = C::kCT;
  tl_gdouble* const W = (tl_gdouble*)W_generic;
  tl_gdouble* const F = (tl_gdouble*)F_generic;
  const double* const l11 = dyn + 64 * C::kLDP;
  const double* const invd = dyn + tl_invd<RS>();
  const int tid = threadIdx.x, wave = tl_uni(tid >> 6), lane = tid & 63;
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  const int nrt = (m2 + 15) / 16;
  const int ntc = (nct - jb - kTlPW) / 16;
  // A wavefront's share is ceil(ntc / NW) column tiles, taken <= CT at a time: the passes are cut in whole tiles PER WAVEFRONT (28 tiles, 4 wavefronts, CT = 3: passes of
  // 12, 8, 8 tiles = 3 + 2 + 2 per wavefront; cut evenly — 10, 9, 9 — every pass costs its busiest wavefront 3)
  const int units = (ntc + NW - 1) / NW;
  const int npass = (units + CT - 1) / CT;
  int t0 = 0, units_left = units;
  for (int pass = 0; pass < npass; ++pass) {
    const int cu = (units_left + (npass - pass) - 1) / (npass - pass);
    units_left -= cu;
    const int tp = (ntc - t0) < cu * NW ? (ntc - t0) : cu * NW;  // column tiles of this pass (<= NW CT)
    // tiles per wavefront: the choice that leaves the busiest wavefront the least work (row tiles x column tiles + its U12)
    int cpw = 1, best = 0x7fffffff;
#pragma unroll
    for (int c = 1; c <= CT; ++c) {
      const int g = (tp + c - 1) / c;
      if (g > NW) continue;
      const int rsp = NW / g;
      const int cost = ((nrt + rsp - 1) / rsp) * 2 * c + 5 * c;
      if (cost <= best) { best = cost; cpw = c; }
    }
    const int ncg = (tp + cpw - 1) / cpw, rsplit = NW / ncg;
    const int cg = wave % ncg, rs = wave / ncg;
    const bool active = rs < rsplit;
    const int tfirst = t0 + cg * cpw;
    const int mine = !active ? 0 : ((tp - cg * cpw) < cpw ? (tp - cg * cpw) : cpw);
    const int c_first = jb + kTlPW + 16 * tfirst;
    // every wavefront takes part in the staging of L21 and its barriers, with or without columns of its own
    switch (mine) {
      case 0: tl_colgroup<RS, 1>(W, F, dyn, l11, invd, s_prow, s_rowlist, n, ldw, jb, c_first, rs, rsplit, nrt, m2, wave, lane, false); break;
      case 1: tl_colgroup<RS, 1>(W, F, dyn, l11, invd, s_prow, s_rowlist, n, ldw, jb, c_first, rs, rsplit, nrt, m2, wave, lane, true); break;
      case 2: tl_colgroup<RS, 2>(W, F, dyn, l11, invd, s_prow, s_rowlist, n, ldw, jb, c_first, rs, rsplit, nrt, m2, wave, lane, true); break;
      case 3: tl_colgroup<RS, 3>(W, F, dyn, l11, invd, s_prow, s_rowlist, n, ldw, jb, c_first, rs, rsplit, nrt, m2, wave, lane, true); break;
      default:
        if constexpr (CT >= 4) tl_colgroup<RS, 4>(W, F, dyn, l11, invd, s_prow, s_rowlist, n, ldw, jb, c_first, rs, rsplit, nrt, m2, wave, lane, true);
        break;
    }
    t0 += tp;
  }
  __syncthreads();  // the tiles stored above are read by other wavefronts from here on
  if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[3] += now - tprev; }
}

// RS = rows per lane of the panel's register layout: 8 for n <= 512 (two workgroups per CU), 16 for n <= 1024
template <int RS>
__global__ __launch_bounds__(64 * tl_cfg<RS>::kWaves, 2) void k_lu_factor_tiled(int n, int ldw, double* __restrict__ w_all, double* __restrict__ f_all, int32_t* __restrict__ piv_all,
                                                                 unsigned long long* singular_word, unsigned int epoch, unsigned long long* phase_clocks) {
  using C = tl_cfg<RS>;
  constexpr int NW = C::kWaves, NT = 64 * NW;
  constexpr int R = RS / NW;
  constexpr int MAXN = C::kMaxN;
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  auto mark = [&](int phase) {
    if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[phase] += now - tprev; tprev = now; }
  };
  extern __shared__ double dyn[];                // see tiled_lds_doubles
  double* const l11 = dyn + 64 * C::kLDP;
  __shared__ short s_pos[MAXN], s_rowat[MAXN];  // position of every row under the reference's interchanges (-1: no such row) and its inverse
  __shared__ int s_prow[kTlPW], s_ipiv[kTlPW];        // the panel's pivot rows (row indices) and recorded pivots (positions)
  __shared__ unsigned short s_rowlist[MAXN + 16];
  __shared__ int s_wcnt[R][NW];
  __shared__ int s_hdr[8], s_flags[1];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tl_stagger_ticks > 0 && (blockIdx.x & 3) != 0 && blockIdx.x < 256) {  // the first round of workgroups only: the ones that follow inherit their CU's phase
    const unsigned long long t0s = wall_clock64(), waits = (unsigned long long)(blockIdx.x & 3) * (unsigned long long)tl_stagger_ticks;
    while (wall_clock64() - t0s < waits) __builtin_amdgcn_s_sleep(64);
  }
  double* const W = w_all + (size_t)blockIdx.x * n * ldw;
  double* const F = f_all + (size_t)blockIdx.x * n * n;
  int32_t* const PIV = piv_all + (size_t)blockIdx.x * n;
  for (int r = tid; r < MAXN; r += NT) { s_pos[r] = (short)(r < n ? r : -1); s_rowat[r] = (short)r; }
  for (int r = tid; r < n + 16; r += NT) s_rowlist[r] = (unsigned short)(r < n ? r : n - 1);  // rows entering the first panel (padded: whole tiles)
  int m_act = n;
  if (tid == 0) s_flags[0] = 0;
  __syncthreads();
  const int nct = (n + 15) / 16 * 16;  // columns processed by the tile phases (the padding up to it stays isolated in its own columns)

  for (int jb = 0; jb < n; jb += kTlPW) {
    const int pw = (n - jb) < kTlPW ? (n - jb) : kTlPW;
    // =========================================================== panel: sub-panels of kSW columns in registers
    tl_panel<RS>(W, ldw, n, jb, dyn, s_pos, s_rowat, s_prow, s_ipiv, s_hdr, s_flags, s_rowlist, m_act, phase_clocks);
    mark(0);
    // =========================================================== the 64 finished rows -> F; list of the rows still active; L11
    int m2 = 0;
    {
      bool act[R];
      unsigned long long bal[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int row = tid + NT * i;
        act[i] = row < n && s_pos[row] >= jb + pw;
        bal[i] = __ballot(act[i]);
        if (lane == 0) s_wcnt[i][wave] = __popcll(bal[i]);
      }
      __syncthreads();
      int base = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        int before = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) { const int cnt = s_wcnt[i][w2]; total += cnt; if (w2 < wave) before += cnt; }
        if (act[i]) s_rowlist[base + before + __popcll(bal[i] & ((1ull << lane) - 1ull))] = (unsigned short)(tid + NT * i);
        base += total;
      }
      m2 = base;
    }
    m_act = m2;
    const int mc = nct - jb - kTlPW;  // trailing columns (exist only behind a full panel)
    const bool trailing = pw == kTlPW && mc > 0 && m2 > 0;
    if (trailing) {
      // the blocks of L11 below its diagonal blocks (rows 16..63, columns 0..47); the inverses of the diagonal blocks are in LDS already (tl_invert_diag)
      for (int idx = tid; idx < 48 * 48; idx += NT) {
        const int k = 16 + idx / 48, i = idx % 48;
        l11[(k - 16) * kTlL11P + i] = i < k ? W[(size_t)s_prow[k] * ldw + jb + i] : (i == k ? 1.0 : 0.0);
      }
    }
#if DSH_TL_FINISH_T
    // the finished rows' entries in the columns 0 .. jb + pw - 1 (their L part and U11) to F, 64 columns at a time through a [64][65] tile in LDS (the panel's
    // buffers are dead, l11 and invd lie behind it): a row is read as 512 adjacent bytes and a column of F is written as 512 adjacent bytes — the direct form
    // (a thread per column, 16 stores of one double each to 64 different lines per wavefront) was bound by the number of lines its stores touch
    {
      double* const tile = dyn;
      constexpr int KPW = kTlPW / NW;  // rows / columns of a block per wavefront
      for (int cb0 = 0; cb0 < jb + pw; cb0 += 64) {
        const bool incol = cb0 + lane < jb + pw;
        double v[KPW];
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          const int k = wave + NW * i;
          v[i] = (k < pw && incol) ? W[(size_t)s_prow[k] * ldw + cb0 + lane] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < KPW; ++i) tile[(wave + NW * i) * 65 + lane] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          const int cc = cb0 + wave + NW * i;
          if (cc < jb + pw && lane < pw) F[(size_t)cc * n + jb + lane] = tile[lane * 65 + wave + NW * i];
        }
        __syncthreads();
      }
    }
#else
    for (int c = tid; c < jb + pw; c += NT) {
      double* const dst = F + (size_t)c * n + jb;
      for (int k0 = 0; k0 < pw; k0 += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (k0 + k < pw) ? W[(size_t)s_prow[k0 + k] * ldw + c] : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k0 + k < pw) dst[k0 + k] = v[k];
      }
    }
#endif
    if (tid < pw) PIV[jb + tid] = s_ipiv[tid];
    __syncthreads();
    if (m2 > 0 && tid < 16) s_rowlist[m2 + tid] = s_rowlist[m2 - 1];  // padding of the last row tile: a valid row, never stored
    if (!trailing) { mark(1); continue; }
    __syncthreads();
    mark(1);
#if DSH_TL_COLS
    tl_trailing_cols<RS>(W, F, dyn, s_prow, s_rowlist, phase_clocks, n, ldw, jb, nct, m2);
#else
    tl_trailing<RS>(W, F, dyn, s_prow, s_rowlist, phase_clocks, n, ldw, jb, nct, m2);
#endif
    if (prof) tprev = wall_clock64();
  }
  if (tid == 0 && s_flags[0] != 0) publish_singular(singular_word, 1ull, epoch);
}

