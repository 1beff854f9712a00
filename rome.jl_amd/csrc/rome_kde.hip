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
// Fast path, N <= 128 (two particles per lane): the same golden-section search -- bracket arithmetic in double, identical iterates --
// with the likelihood evaluated in single precision on the hardware exponential:
//   * row sums: particle i adds exp2(a2 d_ij²) over the N-1 others, j = i+1 .. i+N-1 (mod N), read from an array of partner PAIRS
//     in LDS, q[m] = (y[m mod N], y[(m+64) mod N]): the partners of a lane's two particles at ring distance k are the one 8-byte
//     word q[lane + k] -- no index wrap, lane-consecutive addresses, constant offsets; the j = i term never occurs, so an isolated
//     particle's tiny sum is not cancelled against a self term.  Per term: f32 subtract, two multiplies, v_exp_f32, accumulate
//     (8-term single-precision partials folded into double sums); no exchange of symmetric terms, no fences.  Issue cost on
//     gfx950 (scripts/ubench/ubench32: plain f32 VALU 2.5 cycles per wave, packed f32 5.0 -- the same rate per element --
//     v_exp_f32 8.25): 40 cycles per term of two pairs, 16.5 of them the two exponentials;
//     The particles are staged as single-precision offsets from particle 0 (a fixed 6e-8·range perturbation of the data, the same
//     for every h);
//   * Euclidean stopping rules (>= 1e-2): a golden-section step only COMPARES two likelihoods; when they differ by less than
//     kTieEps (30x the single-precision noise of a likelihood) both are re-evaluated in double precision (lcv_negll), so every
//     decision is the double-precision one and the iterates are the oracle's;
//   * finer stopping rules (the 1e-6 of circular coordinates): the golden section stops at a 1e-2 bracket and the search finishes
//     on the DERIVATIVE, g(h) = Σ_i T_i/S_i − N h² = 0 with T_i = Σ_j w_ij d_ij² (accumulated in the same pass), by secant steps.
//     The zero of g is conditioned like 1e-7/curvature in single precision, whereas comparing likelihood VALUES 1e-6 apart needs
//     double precision: the 33 double-precision evaluations of a heading become ~15 + 3 single-precision ones, and a near-tie
//     decision of the golden section cannot hurt (the secant may leave the last bracket by its width).
// Measured (scripts/kde_profile.py, profiles/r02_kde_bandwidth.txt).
constexpr double kTieEps = 3e-5;
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool CIRC, bool WITH_T>
__device__ __forceinline__ void lcv_eval32(const f32x2 xi, const bool act0, const bool act1, const f32x2* __restrict__ q, int N, double h,
                                           double* negll, double* g) {
  // q: the wave's pair array + lane: q[k] = (particle lane + k, particle lane + 64 + k), indices modulo N -- the partners of the
  // lane's two particles at ring distance k = 1 .. N-1, ONE 8-byte read per term.  WITH_T: also T_i = Σ_j w_ij d_ij² (the
  // derivative g; only the finish of the fine stopping rules reads it: 5 of the 45 issue cycles of a term)
  const float a2 = (float)(-0.72134752044448170368 / (h * h));   // -½ log2(e) / h²
  double S0 = 0.0, S1 = 0.0;
  f32x2 T = {0.0f, 0.0f};
  auto term = [&](const f32x2 xj, f32x2& ps) {
    f32x2 d = xi - xj;
    if (CIRC) {
      d.x = fmaf(-6.2831853071795865f, rintf(d.x * 0.15915494309189535f), d.x);
      d.y = fmaf(-6.2831853071795865f, rintf(d.y * 0.15915494309189535f), d.y);
    }
    const f32x2 d2 = d * d;
    f32x2 w = d2 * a2;
    w.x = __builtin_amdgcn_exp2f(w.x); w.y = __builtin_amdgcn_exp2f(w.y);
    ps += w;
    if (WITH_T) T = __builtin_elementwise_fma(w, d2, T);
  };
  const int M = N - 1, M8 = M & ~7;
  int k = 1;
  for (; k <= M8; k += 8) {
    f32x2 ps = {0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 8; ++u) term(q[k + u], ps);
    S0 += (double)ps.x; S1 += (double)ps.y;
  }
  {
    f32x2 ps = {0.0f, 0.0f};
    for (; k <= M; ++k) term(q[k], ps);
    S0 += (double)ps.x; S1 += (double)ps.y;
  }
  double ll = 0.0, gg = 0.0;
  if (act0) { const double s = fmax(S0, 1e-300); ll += fast_log(s); if (WITH_T) gg += (double)T.x / s; }
  if (act1) { const double s = fmax(S1, 1e-300); ll += fast_log(s); if (WITH_T) gg += (double)T.y / s; }
  if (WITH_T) {
    double v[2] = {ll, gg};
    wave_sum_n<2>(v);
    ll = v[0]; *g = v[1] - (double)N * h * h;
  } else { ll = wave_sum(ll); *g = 0.0; }
  *negll = -(ll - (double)N * fast_log((double)(N - 1) * h * 2.50662827463100050241576528));
}

// g(h) = Σ_i T_i/S_i − N h² in double precision (row sums over the double-precision particles in LDS, j = i skipped): ONE such
// evaluation polishes the single-precision secant result -- for a flat likelihood the single-precision zero of g is only good to
// ~1e-4, the slope is not the problem
template <bool CIRC>
__device__ __forceinline__ double lcv_g64(const double (&x)[2], const bool (&act)[2], const double* __restrict__ pts, int N, int lane, double h) {
  const double a = -0.5 / (h * h);
  double S[2] = {0.0, 0.0}, T[2] = {0.0, 0.0};
  for (int j = 0; j < N; ++j) {
    const double xj = pts[j];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double d = x[s] - xj;
      if (CIRC) d = lcv_wrap(d);
      const double d2 = d * d;
      const double w = (lane + 64 * s == j) ? 0.0 : fast_exp_neg(a * d2);
      S[s] += w; T[s] = fma(w, d2, T[s]);
    }
  }
  double gg = 0.0;
#pragma unroll
  for (int s = 0; s < 2; ++s) if (act[s]) gg += T[s] / fmax(S[s], 1e-300);
  return wave_sum(gg) - (double)N * h * h;
}

template <bool CIRC>
__device__ __forceinline__ double lcv_golden_fast(const double (&x)[2], const bool (&act)[2], const double* __restrict__ pts,
                                                  float* __restrict__ p32d, double* __restrict__ wbuf, int N, int lane, double tol, int* n_evals) {
  // bracket: smallest pair distance and extent about particle 0 (oracle: ro_kde_bandwidth_lcv) -- in double, as the slow path
  const double x0v = pts[0];
  double mn = __builtin_inf(), ylo = 0.0, yhi = 0.0;
  f32x2 xi;
  {
    double y0 = x[0] - x0v, y1 = x[1] - x0v;
    if (CIRC) { y0 = lcv_wrap(y0); y1 = lcv_wrap(y1); }
    ylo = fmin(fmin(ylo, y0), y1); yhi = fmax(fmax(yhi, y0), y1);   // idle slots hold particle 0: y = 0
    xi.x = (float)y0; xi.y = (float)y1;
    p32d[384 + lane] = xi.x;                       // (64 < N: every lane has a first particle)
    if (act[1]) p32d[384 + lane + 64] = xi.y;
  }
  for (int j = 0; j < N; ++j) {
    const double xj = pts[j];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      double d = x[s] - xj;
      if (CIRC) d = lcv_wrap(d);
      if (act[s] && lane + 64 * s != j) mn = fmin(mn, fabs(d));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  f32x2* qa = reinterpret_cast<f32x2*>(p32d);     // pair array: qa[m] = (y[m mod N], y[(m + 64) mod N]), m < 192
#pragma unroll
  for (int r = 0; r < 3; ++r) { const int m = lane + 64 * r; qa[m] = f32x2{p32d[384 + m % N], p32d[384 + (m + 64) % N]}; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  const f32x2* q = qa + lane;
  mn = wave_min(mn); ylo = wave_min(ylo); yhi = -wave_min(-yhi);
  const double minm = fmax(mn, 1e-6), maxm = fmax(yhi - ylo, minm);
  const double ax = 2.0 * minm / (double)(N - 1), bx = 0.5 * (minm + maxm), cx = 2.0 * maxm;
  constexpr double Cg = 0.38196601125010515180, Rg = 0.61803398874989484820;
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + Cg * (cx - bx); }
  else { x2 = bx; x1 = bx - Cg * (bx - ax); }
  double f1, f2, g1, g2;
  const bool finish = tol < 1e-2;          // finer than the golden section is run in single precision: finish on the derivative
  auto eval = [&](double hh, double* f, double* gd) {
    if (finish) lcv_eval32<CIRC, true>(xi, act[0], act[1], q, N, hh, f, gd); else lcv_eval32<CIRC, false>(xi, act[0], act[1], q, N, hh, f, gd);
  };
  eval(x1, &f1, &g1);
  eval(x2, &f2, &g2);
  int ne = 2;
  const double tol_gs = finish ? 1e-2 : tol;
  while (fabs(x3 - x0) > tol_gs * (fabs(x1) + fabs(x2)) && ne < 200) {
    bool lower2 = f2 < f1;
    if (!finish && fabs(f2 - f1) < kTieEps) {   // wave-uniform: decide in double precision
      const double e1 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x1), e2 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x2);
      lower2 = e2 < e1;
    }
    if (lower2) { x0 = x1; x1 = x2; x2 = Rg * x1 + Cg * x3; f1 = f2; g1 = g2; eval(x2, &f2, &g2); }
    else        { x3 = x2; x2 = x1; x1 = Rg * x2 + Cg * x0; f2 = f1; g2 = g1; eval(x1, &f1, &g1); }
    ++ne;
  }
  double best = f1 < f2 ? x1 : x2;
  if (!finish && fabs(f2 - f1) < kTieEps) {
    const double e1 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x1), e2 = lcv_negll<2, CIRC>(x, act, pts, wbuf, N, lane, x2);
    best = e1 < e2 ? x1 : x2;
  }
  if (finish) {
    // secant steps on g from the two interior points; the iterate may leave the last bracket by its width (a near-tie decision
    // of the single-precision golden section), never further (flat or noisy g: keep the golden-section answer)
    const double wdt = x3 - x0, lo = fmax(x0 - wdt, 0.5 * x0), hi = x3 + wdt;
    double ha = x1, ga = g1, hb = x2, gb = g2;
    for (int it = 0; it < 5; ++it) {
      const double den = gb - ga;
      if (!(fabs(den) > 0.0)) break;
      const double hc = hb - gb * (hb - ha) / den;
      if (!(hc > lo && hc < hi)) break;
      double fc, gc;
      eval(hc, &fc, &gc);
      ++ne;
      const bool done = fabs(hc - hb) <= tol * fabs(hc);
      ha = hb; ga = gb; hb = hc; gb = gc;
      best = hc;
      if (done) break;
    }
    if (best == hb && fabs(gb - ga) > 0.0 && hb != ha) {   // one Newton step with the double-precision value of g and the secant slope
      const double g64 = lcv_g64<CIRC>(x, act, pts, N, lane, hb);
      const double hn = hb - g64 * (hb - ha) / (gb - ga);
      ++ne;
      if (hn > lo && hn < hi && fabs(hn - hb) < 1e-2 * hb) best = hn;
    }
  }
  *n_evals = ne;
  return best;
}

template <int S>
__global__ void __launch_bounds__(64 * kKdeWaves) k_kde_bandwidth(int T, int dim, int N, const double* __restrict__ bel,
                                                                  uint32_t circ_mask, double tol_e, double tol_c,
                                                                  double* __restrict__ bw, int32_t* __restrict__ evals) {
  __shared__ double pts[kKdeWaves][64 * S];
  __shared__ double wex[kKdeWaves][64 * S];   // per-wave exchange row of the symmetric likelihood evaluation
  __shared__ __align__(8) float p32buf[kKdeWaves][S == 2 ? 512 : 2];   // fast path: 192 partner pairs (single-precision offsets from particle 0) | the 128 offsets
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int t = blockIdx.x * kKdeWaves + wave;
  if (t >= T) return;   // wave-uniform; nothing below synchronises across waves
  const bool circ = (circ_mask >> (t % dim)) & 1u;
  const double* __restrict__ P = bel + (size_t)t * N;
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
  double h;
  if constexpr (S == 2) {
    float* p32 = reinterpret_cast<float*>(p32buf[wave]);
    h = circ ? lcv_golden_fast<true>(x, act, pts[wave], p32, wex[wave], N, lane, tol_c, &ne)
             : lcv_golden_fast<false>(x, act, pts[wave], p32, wex[wave], N, lane, tol_e, &ne);
  } else {
    h = circ ? lcv_golden<S, true>(x, act, pts[wave], wex[wave], N, lane, tol_c, &ne)
             : lcv_golden<S, false>(x, act, pts[wave], wex[wave], N, lane, tol_e, &ne);
  }
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
                                double* bw, int32_t* evals, hipStream_t s) {
  const int T = V * dim;
  if (T <= 0) return hipSuccess;
  const dim3 grid((T + kKdeWaves - 1) / kKdeWaves), block(64 * kKdeWaves);
#define ROME_LAUNCH_KDE(SS) hipLaunchKernelGGL((k_kde_bandwidth<SS>), grid, block, 0, s, T, dim, N, bel, circ_mask, tol_e, tol_c, bw, evals)
  if (N <= 64) ROME_LAUNCH_KDE(1);
  else if (N <= 128) ROME_LAUNCH_KDE(2);
  else if (N <= 256) ROME_LAUNCH_KDE(4);
  else ROME_LAUNCH_KDE(8);
#undef ROME_LAUNCH_KDE
  return hipGetLastError();
}

}  // namespace rome
