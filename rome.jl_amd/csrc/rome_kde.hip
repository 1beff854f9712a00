// rome_kde.hip -- KDE bandwidths by leave-one-out likelihood cross-validation.
//
// The reference wraps every convolution result in `manikde!` (⚠AMP), which selects one bandwidth per coordinate by
// maximising the leave-one-out log-likelihood of a 1-D Gaussian KDE (Euclidean coordinates: ⚠KDE.jl ksize "lcv",
// golden section to 1 %; Circular coordinates: naive cross-validation with Optim's GoldenSection).  The rule is the
// one written down in oracle/rome_oracle.c (ro_kde_bandwidth_lcv), which reproduces the 361 x 3 bandwidths stored
// with the reference's solved Manhattan-500 graph (tests/golden/manhattan500_reference_solve.npz).
//
// Kernel shape: one wave per (belief, coordinate) task, four tasks per 256-thread block.  Lane l owns particles
// l, l+64, ... (S slots, N <= 64 S); the N kernel points are staged once in the wave's own LDS region and walked with
// reads.  One likelihood evaluation is ⌊N/2⌋ x S x (difference, square, 20-instruction exp) per lane -- every kernel weight
// is evaluated once and exchanged through LDS (lcv_negll); the golden-section search (17 evaluations at 1 %, ~33 at 1e-6) runs on wave-uniform scalars, so there is no
// divergence and no block-level synchronisation.  Compute-bound: 8 N bytes in, 8 bytes out per task.
#include <type_traits>
#include "rome_device_math.hpp"
#include "rome_kernels.h"

namespace rome {

constexpr int kKdeWaves = 4;

__device__ __forceinline__ double lcv_wrap(double d) { return d - 6.283185307179586476925287 * rint(d * 0.15915494309189533576888); }

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
  return v;
}

// -LL(h) of the wave's task; x[s] = own particles (idle slots shadow particle 0), pts = all N particles in LDS.
// Every kernel weight w_ij = w_ji is evaluated ONCE: for the offsets k = 1 .. ⌊N/2⌋ particle i evaluates w(i, i+k mod N), adds it to its
// own sum and publishes it in the wave's LDS exchange row, from which particle i+k picks it up (w(i, i+k) is its "backward" term);
// for even N the offset N/2 pairs i with i+N/2 from both sides, so that round is not exchanged.  N-1 terms per particle from ⌊N/2⌋
// exponentials instead of N-1: ≈ 1.6x fewer VALU instructions per likelihood evaluation than the plain double loop.
template <int S, bool CIRC>
__device__ __forceinline__ double lcv_negll(const double (&x)[S], const bool (&act)[S], const double* __restrict__ pts,
                                            double* __restrict__ wbuf, int N, int lane, double h) {
  const double a = -0.5 / (h * h);
  const int H = N >> 1;
  const bool even = (N & 1) == 0;
  double acc[S];
  int jp[S], jm[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    acc[s] = 0.0;
    const int i = act[s] ? lane + 64 * s : 0;
    jp[s] = i; jm[s] = i;
  }
  for (int k = 1; k <= H; ++k) {
    double w[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      jp[s] = jp[s] + 1 >= N ? jp[s] + 1 - N : jp[s] + 1;
      jm[s] = jm[s] - 1 < 0 ? jm[s] - 1 + N : jm[s] - 1;
      double d = x[s] - pts[jp[s]];
      if (CIRC) d = lcv_wrap(d);
      w[s] = fast_exp_neg(a * d * d);
      acc[s] += w[s];
    }
    if (!(even && k == H)) {   // wave-uniform
#pragma unroll
      for (int s = 0; s < S; ++s) if (act[s]) wbuf[lane + 64 * s] = w[s];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] += wbuf[jm[s]];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
  }
  double ll = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) if (act[s]) ll += fast_log(fmax(acc[s], 1e-300));
  ll = wave_sum(ll);
  return -(ll - (double)N * fast_log((double)(N - 1) * h * 2.50662827463100050241576528));
}

template <int S, bool CIRC>
__device__ __forceinline__ double lcv_golden(const double (&x)[S], const bool (&act)[S], const double* __restrict__ pts,
                                             double* __restrict__ wbuf, int N, int lane, double tol, int* n_evals) {
  // bracket: smallest pair distance and extent about particle 0 (oracle: ro_kde_bandwidth_lcv)
  const double x0v = pts[0];
  double mn = __builtin_inf(), ylo = 0.0, yhi = 0.0;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    double y = x[s] - x0v;
    if (CIRC) y = lcv_wrap(y);
    ylo = fmin(ylo, y); yhi = fmax(yhi, y);   // idle slots hold particle 0: y = 0
  }
  for (int j = 0; j < N; ++j) {
    const double xj = pts[j];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      double d = x[s] - xj;
      if (CIRC) d = lcv_wrap(d);
      if (act[s] && lane + 64 * s != j) mn = fmin(mn, fabs(d));
    }
  }
  mn = wave_min(mn); ylo = wave_min(ylo); yhi = -wave_min(-yhi);
  const double minm = fmax(mn, 1e-6), maxm = fmax(yhi - ylo, minm);
  const double ax = 2.0 * minm / (double)(N - 1), bx = 0.5 * (minm + maxm), cx = 2.0 * maxm;
  constexpr double Cg = 0.38196601125010515180, Rg = 0.61803398874989484820;
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + Cg * (cx - bx); }
  else { x2 = bx; x1 = bx - Cg * (bx - ax); }
  double f1 = lcv_negll<S, CIRC>(x, act, pts, wbuf, N, lane, x1), f2 = lcv_negll<S, CIRC>(x, act, pts, wbuf, N, lane, x2);
  int ne = 2;
  while (fabs(x3 - x0) > tol * (fabs(x1) + fabs(x2)) && ne < 200) {
    if (f2 < f1) { x0 = x1; x1 = x2; x2 = Rg * x1 + Cg * x3; f1 = f2; f2 = lcv_negll<S, CIRC>(x, act, pts, wbuf, N, lane, x2); }
    else         { x3 = x2; x2 = x1; x1 = Rg * x2 + Cg * x0; f2 = f1; f1 = lcv_negll<S, CIRC>(x, act, pts, wbuf, N, lane, x1); }
    ++ne;
  }
  *n_evals = ne;
  return f1 < f2 ? x1 : x2;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Fast path, 8 <= N <= 128 (k_kde_bandwidth_fast): the same golden-section search -- bracket arithmetic in double, identical
// iterates -- with the likelihood evaluated in single precision on the hardware exponential:
//   * row sums by the blocked symmetric scheme below (lcv_eval_blk): every unordered pair of particles is evaluated once.  The
//     particles are staged as single-precision offsets from particle 0 (a fixed 6e-8·range perturbation of the data, the same for
//     every h).  Issue costs on gfx950 (scripts/ubench/ubench32, profiles/r02_ubench_f32_issue.txt): plain f32 VALU 2.5 cycles per
//     wave64, packed f32 5.0 (the same rate per element), v_exp_f32 8.25, f64 fma 5.8;
//   * Euclidean stopping rules (>= 1e-2): a golden-section step only COMPARES two likelihoods; when they differ by less than
//     kTieEps (30x the single-precision noise of a likelihood) both are re-evaluated in double precision (lcv_negll), so every
//     decision is the double-precision one and the iterates are the oracle's;
//   * finer stopping rules (the 1e-6 of circular coordinates): the golden section stops at a 1e-2 bracket and the search finishes
//     on the DERIVATIVE, g(h) = Σ_i T_i/S_i − N h² = 0 with T_i = Σ_j w_ij d_ij² (accumulated in the same pass): secant steps in
//     single precision down to 1e-4, then ONE Newton step with g and g' evaluated in double precision (lcv_g64).  Comparing
//     likelihood VALUES 1e-6 apart needs double precision throughout: the 33 double-precision evaluations of a heading become
//     ~17 + 2 single-precision ones + 1, and a near-tie decision of the golden section cannot hurt (the secant may leave the last
//     bracket by its width).
// History of the evaluation (proposals of a Manhattan sweep, profiles/r02_kde_bandwidth.txt): double precision with symmetric LDS
// exchange 6.6 ms; single-precision ring over ordered pairs 2.2 -> 1.8 ms; blocked symmetric 1.55 ms.
constexpr double kTieEps = 3e-5;

// g(h) = Σ_i T_i/S_i − N h² AND its derivative in double precision (row sums over the double-precision particles in LDS, j = i
// skipped): ONE such evaluation polishes the single-precision secant result by a true Newton step -- for a flat likelihood the
// single-precision zero of g is only good to ~1e-4, and so is a slope taken from single-precision differences.  With
// S_i = Σ_j w_ij, T_i = Σ_j w_ij d², Q_i = Σ_j w_ij d⁴ and ∂w/∂h = w d²/h³:  g'(h) = Σ_i (Q_i S_i − T_i²)/(h³ S_i²) − 2 N h.
template <bool CIRC>
__device__ __forceinline__ void lcv_g64(const double (&x)[2], const bool (&act)[2], const double* __restrict__ pts, int N, int lane, double h,
                                        double* g, double* dg) {
  // Differences, squares and all sums in double precision; only the kernel weight itself goes through the hardware exponential: the
  // exponent a·d²·log2(e) is formed in double and rounded once to single precision (absolute error <= 2e-6 for every weight that
  // matters, i.e. a relative error of ~1e-6 in w_ij, random over the pairs), which moves the Newton step by ~1e-8 h -- two orders
  // below the 2e-6 at which the reference's stored heading bandwidths are reproduced -- at a third of the cost of a double-precision
  // exponential per ordered pair (0.28 -> 0.13 ms of the 1.5 ms of a Manhattan sweep of proposals).
  const double a = -0.5 / (h * h) * 1.4426950408889634074;
  double S[2] = {0.0, 0.0}, T[2] = {0.0, 0.0}, Q[2] = {0.0, 0.0};
  for (int j = 0; j < N; ++j) {
    const double xj = pts[j];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double d = x[s] - xj;
      if (CIRC) d = lcv_wrap(d);
      const double d2 = d * d;
      const double w = (lane + 64 * s == j) ? 0.0 : (double)__builtin_amdgcn_exp2f((float)(a * d2));
      const double wd = w * d2;
      S[s] += w; T[s] += wd; Q[s] = fma(wd, d2, Q[s]);
    }
  }
  double v[2] = {0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 2; ++s) if (act[s]) {
    const double sv = fmax(S[s], 1e-300), ts = T[s] / sv;
    v[0] += ts; v[1] += Q[s] / sv - ts * ts;
  }
  wave_sum_n<2>(v);
  *g = v[0] - (double)N * h * h;
  *dg = v[1] / (h * h * h) - 2.0 * (double)N * h;
}

// ---- blocked SYMMETRIC evaluation of the row sums (fast path, 8 <= N <= 128) -----------------------------------------------------
// The N particles are cut into nb <= 10 blocks of B (7 for N <= 70, 10 for N <= 100, else 13; the tail padded with far-away points of
// weight 0) and
// every lane owns ONE unordered pair of blocks (a <= b): nb(nb+1)/2 <= 55 lanes busy.  A lane evaluates its B x B weights once and
// accumulates them both ways -- B row partials for block a, B column partials for block b -- in registers, with the 2B particle
// values in registers too: per UNORDERED pair 3 VALU + v_exp_f32 + 2 accumulates = 21 issue cycles (the ring scheme before it: 20
// per ORDERED pair).  Partials go to an nb x nb matrix of cells in LDS, cell (p, q) = the partial sums of the rows of block p
// against block q; row i then sums the nb cells of its block row (B-term single-precision partials, single-precision sums of
// <= 10 of them, converted once).  The derivative sums T_i go through the same cells after the row sums have been read.
// 1 / v for a positive normal double: single-precision seed + two Newton steps (relative error ~1e-16; an IEEE division costs 5x)
__device__ __forceinline__ double rcp_pos_f64(double v) {
  const int e = __builtin_amdgcn_frexp_exp(v);                 // the seed is taken on the mantissa: v may be outside the float range
  const double m = __builtin_amdgcn_frexp_mant(v);
  double r = (double)__builtin_amdgcn_rcpf((float)m);
  r = r * fma(-m, r, 2.0);
  r = r * fma(-m, r, 2.0);
  return __builtin_ldexp(r, -e);
}
// Π over the wavefront of values given as (mantissa in [0.25, 1), exponent): the DPP pattern of wave_sum_n with multiplies; the
// mantissa product of a 16-lane row (>= 2^-32) is renormalised before the rows are folded.  Result in every lane.
__device__ __forceinline__ void wave_prod_frexp(double* mant, int* expo) {
  double m = *mant; int e = *expo;
  m *= dpp_mov<0xB1>(m);  e += __builtin_amdgcn_mov_dpp(e, 0xB1, 0xF, 0xF, true);
  m *= dpp_mov<0x4E>(m);  e += __builtin_amdgcn_mov_dpp(e, 0x4E, 0xF, 0xF, true);
  m *= dpp_mov<0x141>(m); e += __builtin_amdgcn_mov_dpp(e, 0x141, 0xF, 0xF, true);
  m *= dpp_mov<0x140>(m); e += __builtin_amdgcn_mov_dpp(e, 0x140, 0xF, 0xF, true);
  e += __builtin_amdgcn_frexp_exp(m); m = __builtin_amdgcn_frexp_mant(m);
  auto bcast = [&](auto ctrl_tag, auto mask_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(m), CTRL, MASK, 0xF, false);            // rows outside the mask: 1.0
    const int hi = __builtin_amdgcn_update_dpp(0x3FF00000, __double2hiint(m), CTRL, MASK, 0xF, false);
    m *= __hiloint2double(hi, lo);
    e += __builtin_amdgcn_update_dpp(0, e, CTRL, MASK, 0xF, false);
  };
  bcast(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});   // row_bcast:15 -> rows 1, 3
  bcast(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});   // row_bcast:31 -> rows 2, 3
  *mant = readlane_f64(m, 63);
  *expo = __builtin_amdgcn_readlane(e, 63);
}
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <int B>
__host__ __device__ constexpr int kdeCells(int N) { const int nb = (N + B - 1) / B; return nb * B * 12 > nb * nb * B ? nb * B * 12 : nb * nb * B; }   // floats per wave (either cell layout)
template <int B>
struct BlkPlan { int nb, a, b; bool ok, diag; };
template <int B>
__device__ __forceinline__ BlkPlan<B> blk_plan(int N, int lane) {
  BlkPlan<B> p;
  p.nb = (N + B - 1) / B;
  p.ok = lane < p.nb * (p.nb + 1) / 2;
  int a = 0, rem = p.ok ? lane : 0;   // idle lanes recompute pair 0 and store nothing
  while (rem >= p.nb - a) { rem -= p.nb - a; ++a; }
  p.a = a; p.b = a + rem; p.diag = rem == 0;
  return p;
}

// The tail of one evaluation: partial sums -> LDS cells -> row sums -> -LL (and g).  Cell layout (ROME_KDE_ROWCELLS, default):
// row-major, cell(i, q) = partial sum of particle i against block q at M[i * kKdeRowPitch + q] -- a row's <= 10 partials are 40
// contiguous bytes, read back with three 16-byte LDS reads in flight (the block-major layout before it walked them with a loop of
// dependent 4-byte reads: ten LDS round trips per evaluation on the critical path of a search that is sequential by nature).
// Columns q >= nb are zeroed once per task (lcv_cells_init) and never written: the sums keep their order and their bits.
#ifndef ROME_KDE_ROWCELLS
#define ROME_KDE_ROWCELLS 1
#endif
constexpr int kKdeRowPitch = 12;   // floats per row of cells (10 used; 48 bytes: 16-byte aligned rows)
template <int B>
__device__ __forceinline__ void lcv_cells_init(float* __restrict__ M, int N, int lane) {
#if ROME_KDE_ROWCELLS
  const int nb = (N + B - 1) / B;
  for (int q = lane; q < nb * B * kKdeRowPitch; q += 64) M[q] = 0.0f;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#endif
}
template <bool WITH_T, int B, bool FOLD64 = WITH_T>
__device__ __forceinline__ void lcv_tail(const BlkPlan<B>& pl, const float (&r)[B], const float (&c)[B], const float (&tr)[B], const float (&tc)[B],
                                         float* __restrict__ M, int N, int lane, double h, double tscale, double* negll, double* g) {
  const int nb = pl.nb;
  const int i0 = lane, i1 = lane + 64;
  const bool act0 = i0 < N, act1 = i1 < N;
#if ROME_KDE_ROWCELLS
  float* cr = M + (pl.a * B) * kKdeRowPitch + pl.b;
  float* cc = M + (pl.b * B) * kKdeRowPitch + pl.a;
  const float* row0 = M + i0 * kKdeRowPitch;
  const float* row1 = M + (act1 ? i1 : 0) * kKdeRowPitch;
  auto exchange = [&](const float (&pr)[B], const float (&pc)[B], double* o0, double* o1) {
    if (pl.ok) {
#pragma unroll
      for (int u = 0; u < B; ++u) cr[u * kKdeRowPitch] = pr[u];
      if (!pl.diag) {
#pragma unroll
        for (int u = 0; u < B; ++u) cc[u * kKdeRowPitch] = pc[u];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    float v0[12], v1[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(row0 + 4 * q), b = *reinterpret_cast<const float4*>(row1 + 4 * q);
      v0[4 * q] = a.x; v0[4 * q + 1] = a.y; v0[4 * q + 2] = a.z; v0[4 * q + 3] = a.w;
      v1[4 * q] = b.x; v1[4 * q + 1] = b.y; v1[4 * q + 2] = b.z; v1[4 * q + 3] = b.w;
    }
    if (FOLD64) {   // the derivative finish resolves 1e-6: the <= 10 partials of a row are folded in double (as the ring scheme did)
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < 10; ++k) { s0 += (double)v0[k]; s1 += (double)v1[k]; }
      *o0 = s0; *o1 = s1;
    } else {
      float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
      for (int k = 0; k < 10; ++k) { s0 += v0[k]; s1 += v1[k]; }
      *o0 = (double)s0; *o1 = (double)s1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // the cells are rewritten next
  };
#else
  float* cr = M + (pl.a * nb + pl.b) * B;
  float* cc = M + (pl.b * nb + pl.a) * B;
  const int b0 = i0 / B, b1 = (act1 ? i1 : 0) / B;
  const float* row0 = M + b0 * nb * B + (i0 - b0 * B);
  const float* row1 = M + b1 * nb * B + ((act1 ? i1 : 0) - b1 * B);
  auto exchange = [&](const float (&pr)[B], const float (&pc)[B], double* o0, double* o1) {
    if (pl.ok) {
#pragma unroll
      for (int u = 0; u < B; ++u) cr[u] = pr[u];
      if (!pl.diag) {
#pragma unroll
        for (int u = 0; u < B; ++u) cc[u] = pc[u];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    if (FOLD64) {
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < nb; ++k) { s0 += (double)row0[k * B]; s1 += (double)row1[k * B]; }
      *o0 = s0; *o1 = s1;
    } else {
      float s0 = 0.0f, s1 = 0.0f;
      for (int k = 0; k < nb; ++k) { s0 += row0[k * B]; s1 += row1[k * B]; }
      *o0 = (double)s0; *o1 = (double)s1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // the cells are rewritten next
  };
#endif
  double S0, S1, T0 = 0.0, T1 = 0.0;
  exchange(r, c, &S0, &S1);
  if (WITH_T) exchange(tr, tc, &T0, &T1);
  // Σ_i log S_i = log Π_i S_i: mantissas multiplied, exponents added (wave_prod_frexp), ONE logarithm per evaluation instead of
  // two per lane
  const double s0 = act0 ? fmax(S0, 1e-300) : 1.0, s1 = act1 ? fmax(S1, 1e-300) : 1.0;
  double mant = __builtin_amdgcn_frexp_mant(s0) * __builtin_amdgcn_frexp_mant(s1);   // in [0.25, 1)
  int expo = __builtin_amdgcn_frexp_exp(s0) + __builtin_amdgcn_frexp_exp(s1);
  double gg = 0.0;
  if (WITH_T) gg = tscale * ((act0 ? T0 * rcp_pos_f64(s0) : 0.0) + (act1 ? T1 * rcp_pos_f64(s1) : 0.0));
  wave_prod_frexp(&mant, &expo);
  // the evaluation's two logarithms -- of the mantissa product and of (N-1) h sqrt(2π), both wave-uniform -- in ONE pass of the
  // logarithm kernel: even lanes take the first argument, odd lanes the second (the same function of the same argument: same bits)
  const double lg = fast_log((lane & 1) ? (double)(N - 1) * h * 2.50662827463100050241576528 : mant);
  const double ll = readlane_f64(lg, 0) + (double)expo * 0.693147180559945309417;
  if (WITH_T) *g = uniform_f64(wave_sum(gg) - (double)N * h * h); else *g = 0.0;
  // (wave-uniform by construction; saying so lets the search state live in scalar registers across the unrolled block body)
  *negll = uniform_f64(-(ll - (double)N * readlane_f64(lg, 1)));
}

template <bool CIRC, bool WITH_T, int B, bool FOLD64 = WITH_T>
__device__ __forceinline__ void lcv_eval_blk(const BlkPlan<B>& pl, const float* __restrict__ xs, float* __restrict__ M, int N, int lane,
                                             double h, double* negll, double* g) {
  const float hf = (float)h;
  const float a2 = -0.72134752f / (hf * hf);   // -½ log2(e) / h²
  float xi[B], xj[B], r[B], c[B];
  [[maybe_unused]] float tr[B], tc[B];
#pragma unroll
  for (int u = 0; u < B; ++u) {
    xi[u] = xs[pl.a * B + u]; xj[u] = xs[pl.b * B + u]; r[u] = 0.0f; c[u] = 0.0f;
    if (WITH_T) { tr[u] = 0.0f; tc[u] = 0.0f; }
  }
#pragma unroll
  for (int ii = 0; ii < B; ++ii) {
#pragma unroll
    for (int jj = 0; jj < B; ++jj) {
      float d = xi[ii] - xj[jj];
      if (CIRC) d = fmaf(-6.2831853071795865f, rintf(d * 0.15915494309189535f), d);
      const float d2 = d * d;
      float w = __builtin_amdgcn_exp2f(d2 * a2);
      if (ii == jj) w = pl.diag ? 0.0f : w;      // a block against itself: every ordered pair once, the particle itself never
      r[ii] += w; c[jj] += w;
      if (WITH_T) { const float tw = w * d2; tr[ii] += tw; tc[jj] += tw; }
    }
    // one row of B pairs in flight: left alone, the compiler evaluates all B² weights first and accumulates afterwards (> 400 VGPRs).
    // The empty asm statements pin the row and column sums after this row and make the next row's particle depend on them.
    asm volatile("" : "+v"(r[ii]));
    if (WITH_T) asm volatile("" : "+v"(tr[ii]));
    if (ii + 1 < B) {
#pragma unroll
      for (int u = 0; u < B; ++u) { asm volatile("" : "+v"(c[u])); if (WITH_T) asm volatile("" : "+v"(tc[u])); }
      asm volatile("" : "+v"(xi[ii + 1]));
    }
  }
  if constexpr (!WITH_T) { lcv_tail<false, B, FOLD64>(pl, r, c, r, c, M, N, lane, h, 1.0, negll, g); }
  else { lcv_tail<true, B>(pl, r, c, tr, tc, M, N, lane, h, 1.0, negll, g); }
}

template <bool CIRC, int B>
__device__ __forceinline__ double lcv_golden_fast(const double (&x)[2], const bool (&act)[2], const double* __restrict__ pts,
                                                  float* __restrict__ xs, float* __restrict__ M, double* __restrict__ wbuf, int N, int lane,
                                                  double tol, int* n_evals) {
  // bracket: smallest pair distance and extent about particle 0 (oracle: ro_kde_bandwidth_lcv) -- in double, as the slow path
  const double x0v = pts[0];
  double mn = __builtin_inf(), ylo = 0.0, yhi = 0.0;
  {
    double y0 = x[0] - x0v, y1 = x[1] - x0v;
    if (CIRC) { y0 = lcv_wrap(y0); y1 = lcv_wrap(y1); }
    ylo = fmin(fmin(ylo, y0), y1); yhi = fmax(fmax(yhi, y0), y1);   // idle slots hold particle 0: y = 0
    xs[lane] = act[0] ? (float)y0 : 1.0e18f;
    xs[lane + 64] = act[1] ? (float)y1 : 1.0e18f;  // padding: far away, weight exp2(-huge) = 0 against everything
    if (lane < 8) xs[128 + lane] = 1.0e18f;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  const BlkPlan<B> pl = blk_plan<B>(N, lane);
  lcv_cells_init<B>(M, N, lane);
  {   // smallest pair distance, in double on the particles themselves: the lane's block pair, every unordered pair once
    double pi[B];
#pragma unroll
    for (int u = 0; u < B; ++u) pi[u] = pts[min(pl.a * B + u, N - 1)];
#pragma unroll 1
    for (int jj = 0; jj < B; ++jj) {
      const int j = pl.b * B + jj;
      const double pj = pts[min(j, N - 1)];
#pragma unroll
      for (int ii = 0; ii < B; ++ii) {
        double d = pi[ii] - pj;
        if (CIRC) d = lcv_wrap(d);
        const bool use = pl.ok && j < N && pl.a * B + ii < N && !(pl.diag && ii == jj);
        mn = fmin(mn, use ? fabs(d) : __builtin_inf());
      }
    }
  }
  mn = uniform_f64(wave_min(mn)); ylo = uniform_f64(wave_min(ylo)); yhi = uniform_f64(-wave_min(-yhi));
  // the search state is wave-uniform: every update goes through readfirstlane (U) so that it lives in scalar registers across the
  // unrolled block body instead of being spilled around it
#define U(v) uniform_f64(v)
  const double minm = fmax(mn, 1e-6), maxm = fmax(yhi - ylo, minm);
  const double ax = U(2.0 * minm / (double)(N - 1)), bx = U(0.5 * (minm + maxm)), cx = U(2.0 * maxm);
  constexpr double Cg = 0.38196601125010515180, Rg = 0.61803398874989484820;
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = U(bx + Cg * (cx - bx)); }
  else { x2 = bx; x1 = U(bx - Cg * (bx - ax)); }
  const bool finish = tol < 1e-2;          // finer than the golden section is run in single precision: finish on the derivative
  const bool nowrap = CIRC && (yhi - ylo) < 3.0;
  const double tol_gs = finish ? 1e-2 : tol;
  // ONE evaluation site (the unrolled B x B body is a few kB of code per instantiation): a small state machine asks for the next h.
  //   phase 0 / 1: the two interior points; 2: golden section (`upper`: the request replaces x2, else x1); 3: secant on g
  double f1 = 0.0, f2 = 0.0, g1 = 0.0, g2 = 0.0, best = x1;
  double ha = 0.0, ga = 0.0, hb = 0.0, gb = 0.0, lo = 0.0, hi = 0.0;
  int ne = 0, phase = 0, it = 0;
  bool upper = false;
  double hq = x1;
  for (;;) {
    double fv, gv;
    if (finish) {
      // a circular coordinate whose particles all lie within 3 rad of each other: every staged difference has |d| < π, its wrap is
      // the identity (rint(d / 2π) = 0, fma(-2π, 0, d) = d exactly) -- the Euclidean body evaluates the same bits with 3 of 12
      // instructions per pair less (wave-uniform choice; headings of pose beliefs are almost always that concentrated)
      // ... and its golden section needs likelihood VALUES only: the derivative sums T_i (3 of 9 instructions per pair, a second
      // exchange, two reciprocals and a wave sum per evaluation) are left out of those ~17 evaluations -- same row sums, same double
      // fold, same f bits -- and g is evaluated at the two interior points the secant starts from (phases 4, 5: two evaluations more)
      if (CIRC && nowrap) {
        if (phase <= 2) lcv_eval_blk<false, false, B, true>(pl, xs, M, N, lane, hq, &fv, &gv);
        else lcv_eval_blk<false, true, B>(pl, xs, M, N, lane, hq, &fv, &gv);
      } else lcv_eval_blk<CIRC, true, B>(pl, xs, M, N, lane, hq, &fv, &gv);
    } else lcv_eval_blk<CIRC, false, B>(pl, xs, M, N, lane, hq, &fv, &gv);
    ++ne;
    if (phase == 0) { f1 = fv; g1 = gv; hq = x2; phase = 1; continue; }
    if (phase == 1) { f2 = fv; g2 = gv; phase = 2; }
    else if (phase == 2) { if (upper) { f2 = fv; g2 = gv; } else { f1 = fv; g1 = gv; } }
    if (phase == 2) {
      if (fabs(x3 - x0) > tol_gs * (fabs(x1) + fabs(x2)) && ne < 200) {
        bool lower2 = f2 < f1;
#ifndef ROME_KDE_EXPERIMENT_NO_TIES
        if (!finish && fabs(f2 - f1) < kTieEps) {   // wave-uniform: decide in double precision
          const double e1 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x1), e2 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x2);
          lower2 = e2 < e1;
        }
#endif
        if (lower2) { x0 = x1; x1 = x2; x2 = U(Rg * x1 + Cg * x3); f1 = f2; g1 = g2; hq = x2; upper = true; }
        else        { x3 = x2; x2 = x1; x1 = U(Rg * x2 + Cg * x0); f2 = f1; g2 = g1; hq = x1; upper = false; }
        continue;
      }
      best = f1 < f2 ? x1 : x2;
      if (!finish) {
#ifndef ROME_KDE_EXPERIMENT_NO_TIES
        if (fabs(f2 - f1) < kTieEps) {
          const double e1 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x1), e2 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x2);
          best = e1 < e2 ? x1 : x2;
        }
#endif
        break;
      }
      if (CIRC && nowrap) { hq = x1; phase = 4; continue; }   // (the derivative at x1, then at x2)
    }
    if (phase == 4) { g1 = gv; hq = x2; phase = 5; continue; }
    if (phase == 2 || phase == 5) {
      if (phase == 5) g2 = gv;
      // secant steps on g from the two interior points; the iterate may leave the last bracket by its width (a near-tie decision
      // of the single-precision golden section), never further (flat or noisy g: keep the golden-section answer)
      const double wdt = x3 - x0;
      lo = U(fmax(x0 - wdt, 0.5 * x0)); hi = U(x3 + wdt);
      ha = x1; ga = g1; hb = x2; gb = g2;
      phase = 3;
    } else {   // phase 3: the evaluation at hq is in (fv, gv)
      const bool done = fabs(hq - hb) <= fmax(tol, 1e-4) * fabs(hq);   // (the double-precision Newton step below squares what is left)
      ha = hb; ga = gb; hb = hq; gb = gv;
      best = hq;
      if (done || ++it >= 5) break;
    }
    const double den = gb - ga;
    if (!(fabs(den) > 0.0)) break;
    const double hc = U(hb - gb * (hb - ha) / den);
    if (!(hc > lo && hc < hi)) break;
    hq = hc;
  }
#ifndef ROME_KDE_EXPERIMENT_NO_G64
  if (finish && best == hb) {   // one Newton step in double precision (value and derivative of g)
    double g64, dg64;
    lcv_g64<CIRC>(x, act, pts, N, lane, hb, &g64, &dg64);
    const double hn = hb - g64 / dg64;
    ++ne;
    if (dg64 < 0.0 && hn > lo && hn < hi && fabs(hn - hb) < 1e-2 * hb) best = hn;
  }
#endif
#undef U
  *n_evals = ne;
  return best;
}

template <int S>
__global__ void __launch_bounds__(64 * kKdeWaves) k_kde_bandwidth(int T, int dim, int N, const double* __restrict__ bel,
                                                                  uint32_t circ_mask, double tol_e, double tol_c,
                                                                  double* __restrict__ bw, int32_t* __restrict__ evals, const int32_t* __restrict__ block_idx) {
  __shared__ double pts[kKdeWaves][64 * S];
  __shared__ double wex[kKdeWaves][64 * S];   // per-wave exchange row of the symmetric likelihood evaluation
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.x * kKdeWaves + wave;
  if (t >= T) return;   // wave-uniform; nothing below synchronises across waves
  const bool circ = (circ_mask >> (t % dim)) & 1u;
  // (block_idx: belief t / dim lives in block block_idx[t / dim] of `bel` -- a scattered subset of a store; wave-uniform)
  const double* __restrict__ P = bel + (block_idx ? (size_t)block_idx[t / dim] * dim + (size_t)(t % dim) : (size_t)t) * N;
  double x[S];
  bool act[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = lane + 64 * s;
    act[s] = i < N;
    x[s] = P[act[s] ? i : 0];
    if (act[s]) pts[wave][i] = x[s];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  int ne = 0;
  const double h = circ ? lcv_golden<S, true>(x, act, pts[wave], wex[wave], N, lane, tol_c, &ne)
                        : lcv_golden<S, false>(x, act, pts[wave], wex[wave], N, lane, tol_e, &ne);
  if (lane == 0) { bw[t] = h; if (evals) evals[t] = ne; }
}

// the fast path as its own kernel per block size (register allocation is per kernel: the B = 13 bodies must not set the occupancy of
// the N = 100 case); four waves per SIMD (128 VGPRs: measured faster than three without the ~70 spills, which sit in the rare
// double-precision paths) -- left alone the scheduler spreads the unrolled block bodies over > 400 VGPRs
// (tried: the pair exponents -½ log2(e) d² of a task held in registers across its ~20 evaluations -- 419 / 593 VALU instructions per
// evaluation instead of 617 .. 1097, but 231 VGPRs = two waves per SIMD: 1.41 ms against 1.345 ms, profiles/r03_kde_experiments.txt)
template <int B>
__global__ void __launch_bounds__(64 * kKdeWaves) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_kde_bandwidth_fast(int T, int dim, int N, const double* __restrict__ bel, uint32_t circ_mask, double tol_e, double tol_c,
                     double* __restrict__ bw, int32_t* __restrict__ evals, const int32_t* __restrict__ block_idx) {
  __shared__ double pts[kKdeWaves][128];
  __shared__ double wex[kKdeWaves][128];   // exchange row of the double-precision evaluations (tie decisions)
  __shared__ float xsbuf[kKdeWaves][136];  // the particles as single-precision offsets from particle 0 (+ padding)
  extern __shared__ float cellbuf[];       // per wave nb x nb cells of B partial sums (sized by the launcher)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.x * kKdeWaves + wave;
  if (t >= T) return;   // wave-uniform; nothing below synchronises across waves
  const bool circ = (circ_mask >> (t % dim)) & 1u;
  // (block_idx: belief t / dim lives in block block_idx[t / dim] of `bel` -- a scattered subset of a store; wave-uniform)
  const double* __restrict__ P = bel + (block_idx ? (size_t)block_idx[t / dim] * dim + (size_t)(t % dim) : (size_t)t) * N;
  double x[2];
  bool act[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int i = lane + 64 * s;
    act[s] = i < N;
    x[s] = P[act[s] ? i : 0];
    if (act[s]) pts[wave][i] = x[s];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  int ne = 0;
  float* M = cellbuf + wave * kdeCells<B>(N);
  const double h = circ ? lcv_golden_fast<true, B>(x, act, pts[wave], xsbuf[wave], M, wex[wave], N, lane, tol_c, &ne)
                        : lcv_golden_fast<false, B>(x, act, pts[wave], xsbuf[wave], M, wex[wave], N, lane, tol_e, &ne);
  if (lane == 0) { bw[t] = h; if (evals) evals[t] = ne; }
}

// ---- max-density point of a belief per coordinate (⚠IIF getKDEMax: PPE `max`, `suggested` heading) --------------------------
// Rule and pin: oracle/rome_oracle.c ro_kde_max (reproduces the 1083 stored ppe.max values of the reference's solved graph).  One wave
// per (belief, coordinate) task; lane l owns grid points l, l+64, l+128, l+192 (G <= 256); particles staged in the wave's LDS region.
constexpr int kKdeMaxN = 512;   // = ROME_MAX_PARTICLES (include/rome_mi355.h)
constexpr int kKdeGridSlots = 4;

__global__ void __launch_bounds__(64 * kKdeWaves) k_kde_max(int T, int N, int G, double extend, const double* __restrict__ bel,
                                                            const double* __restrict__ bw, double* __restrict__ out) {
  __shared__ double pts[kKdeWaves][kKdeMaxN];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.x * kKdeWaves + wave;
  if (t >= T) return;
  const double* __restrict__ P = bel + (size_t)t * N;
  double lo = __builtin_inf(), nhi = __builtin_inf();
  for (int i = lane; i < N; i += 64) { const double x = P[i]; pts[wave][i] = x; lo = fmin(lo, x); nhi = fmin(nhi, -x); }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  lo = wave_min(lo);
  double hi = -wave_min(nhi);
  const double r = hi - lo; lo -= extend * r; hi += extend * r;
  const double step = (hi - lo) / (double)(G - 1), h = fmax(bw[t], 1e-150), a = -0.5 / (h * h);   // a zero bandwidth (unsolved variable) stays finite
  double X[kKdeGridSlots], y[kKdeGridSlots];
#pragma unroll
  for (int s = 0; s < kKdeGridSlots; ++s) {
    const int g = lane + 64 * s;
    X[s] = g == G - 1 ? hi : lo + (double)g * step;
    y[s] = 0.0;
  }
  for (int j = 0; j < N; ++j) {
    const double xj = pts[wave][j];
#pragma unroll
    for (int s = 0; s < kKdeGridSlots; ++s) { const double d = X[s] - xj; y[s] += fast_exp_neg(a * d * d); }
  }
  // first grid index attaining the maximum: per lane (ascending g), then across lanes (ties -> smaller g)
  double best = -1.0; int gb = 0;
#pragma unroll
  for (int s = 0; s < kKdeGridSlots; ++s) { const int g = lane + 64 * s; if (g < G && y[s] > best) { best = y[s]; gb = g; } }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const double ob = __shfl_xor(best, m, 64); const int og = __shfl_xor(gb, m, 64);
    if (ob > best || (ob == best && og < gb)) { best = ob; gb = og; }
  }
  if (lane == 0) out[t] = gb == G - 1 ? hi : lo + (double)gb * step;
}

hipError_t launch_kde_max(int dim, int V, int N, int G, double extend, const double* bel, const double* bw, double* out, hipStream_t s) {
  const int T = V * dim;
  if (T <= 0) return hipSuccess;
  if (N < 1 || N > kKdeMaxN || G < 2 || G > 64 * kKdeGridSlots) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_kde_max, dim3((T + kKdeWaves - 1) / kKdeWaves), dim3(64 * kKdeWaves), 0, s, T, N, G, extend, bel, bw, out);
  return hipGetLastError();
}

hipError_t launch_kde_bandwidth(int dim, int V, int N, const double* bel, uint32_t circ_mask, double tol_e, double tol_c,
                                double* bw, int32_t* evals, hipStream_t s, const int32_t* block_idx) {
  const int T = V * dim;
  if (T <= 0) return hipSuccess;
  const dim3 grid((T + kKdeWaves - 1) / kKdeWaves), block(64 * kKdeWaves);
#define ROME_LAUNCH_KDE(SS) hipLaunchKernelGGL((k_kde_bandwidth<SS>), grid, block, 0, s, T, dim, N, bel, circ_mask, tol_e, tol_c, bw, evals, block_idx)
  if (N < 8) ROME_LAUNCH_KDE(1);
  else if (N <= 70)
    hipLaunchKernelGGL((k_kde_bandwidth_fast<7>), grid, block, sizeof(float) * kKdeWaves * (size_t)kdeCells<7>(N), s, T, dim, N, bel, circ_mask,
                       tol_e, tol_c, bw, evals, block_idx);
  else if (N <= 100)
    hipLaunchKernelGGL((k_kde_bandwidth_fast<10>), grid, block, sizeof(float) * kKdeWaves * (size_t)kdeCells<10>(N), s, T, dim, N, bel, circ_mask,
                       tol_e, tol_c, bw, evals, block_idx);
  else if (N <= 128)
    hipLaunchKernelGGL((k_kde_bandwidth_fast<13>), grid, block, sizeof(float) * kKdeWaves * (size_t)kdeCells<13>(N), s, T, dim, N, bel, circ_mask,
                       tol_e, tol_c, bw, evals, block_idx);
  else if (N <= 256) ROME_LAUNCH_KDE(4);
  else ROME_LAUNCH_KDE(8);
#undef ROME_LAUNCH_KDE
  return hipGetLastError();
}

}  // namespace rome
