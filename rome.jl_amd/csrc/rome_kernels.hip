// rome_kernels.hip -- batched factor-convolution kernels for gfx950 (MI355X).
//
// One convolution = approxConvBelief for one (factor, direction): N particle root-finds
// (IncrementalInference `computeAcrossHypothesis!` -> `_solveCCWNumeric!`)
// around the RoME residual functors (rows a2/a4/a6/a8).
//
// Two mappings:
//  * k_conv_flat -- the plain whole-graph sweep of a unique-root factor (closed form / Newton, in-kernel noise): the particles of
//    consecutive convolutions are PACKED onto the threads of a 256-thread block (thread = two neighbouring particles, 50 threads
//    per N = 100 convolution, 5 convolutions = 250 of 256 threads), per-factor μ / chol Σ staged through LDS.
//  * k_conv -- ONE WAVEFRONT PER CONVOLUTION for everything that needs a statistic over the N particles of a belief (the inflation
//    spread of the iterative solvers, multihypo / nullhypo) or pre-sampled noise.  Lane l owns particles 2l, 2l+1, 128+2l, ...
//    (PPL per lane, in registers), so
//   * the belief blocks are SoA [var][dim][N]: a wave reads/writes contiguous runs -> coalesced;
//   * per-factor constants (μ, chol Σ, var ids, direction) are wave-uniform -> scalar loads / SGPRs;
//   * the per-cycle belief statistics IIF needs for the entropy inflation (std of the N target
//     points) are pure wave64 xor-butterflies -- no LDS round trip, no __syncthreads();
//   * waves of a workgroup never wait for each other (Nelder-Mead trip counts differ per wave).
// Blocks are 256 threads = 4 convolutions; blockIdx is remapped so that each XCD (own L2) works on
// a contiguous range of the convolution table (neighbouring factors share variables).
#include <cstdlib>
// Floating-point contraction by SOURCE EXPRESSION (a*b + c written in one expression is one fma), not across statements at the
// optimizer's discretion (hipcc's default, -ffp-contract=fast): the same inlined function then rounds identically in every kernel
// instantiation it is inlined into -- the packed sweep, the wave-per-row kernel (lean or not) and the per-factor entry points agree
// bit for bit for every solver (tests/test_gpu_config4.py), which "fast" does not guarantee.
#pragma clang fp contract(on)
#include "rome_device_math.hpp"
#include "rome_kernels.h"

namespace rome {

// block b runs on XCD b % 8 (observed dispatch rule; only used for L2 locality, never correctness).
__device__ __forceinline__ int xcd_contiguous_block(int b, int nb) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = b & 7, idx = b >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ------------------------------------------------------------------------------------------
// factor policies
// ------------------------------------------------------------------------------------------
// Coordinate form of the Pose2Pose2 residual (same function as
// src/factors/Pose2D.jl:51-67 evaluated through exp/compose/log on points):
//   r(z; p, q) = ( p.t + R(θp) z_t - q.t ,  wrap(θp + zθ - θq) )
// dir 0 (solve q): q̂ = p ∘ exp(z) is constant -> r = (q̂.t - q.t, wrap(q̂θ - qθ)): no transcendental per evaluation.
// dir 1 (solve p): one sincos(θp) per evaluation.
struct P2P2Cost {
  double zx, zy, a0, a1, a2; int dir;  // dir0: a = q̂ (x,y,θ) ; dir1: a = (q.x, q.y, qθ - zθ)
  __device__ __forceinline__ double operator()(const double (&x)[3]) const {
    double r0, r1, r2;
    if (dir == 0) { r0 = a0 - x[0]; r1 = a1 - x[1]; r2 = wrap_pi(a2 - x[2]); }
    else {
      double s, c; fast_sincos(x[2], &s, &c);
      r0 = x[0] + c * zx - s * zy - a0; r1 = x[1] + s * zx + c * zy - a1; r2 = wrap_pi(x[2] - a2);
    }
    return r0 * r0 + r1 * r1 + r2 * r2;
  }
};

// std of the belief's tangent coordinates about particle 0 (shifted one-pass moments): SE(2)
template <int PPL>
__device__ __forceinline__ double spread_se2(const double (&t)[PPL][3], const bool (&act)[PPL], double inv, double den) {
  const double x0 = readlane_f64(t[0][0], 0), y0 = readlane_f64(t[0][1], 0), th0 = readlane_f64(t[0][2], 0);
  double s[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const double dx = t[k][0] - x0, dy = t[k][1] - y0, dt = wrap_pi(t[k][2] - th0);
    if (act[k]) { s[0] += dx; s[1] += dy; s[2] += dt; s[3] += dx * dx + dy * dy + dt * dt; }
  }
  // only the SUM of the coordinate variances is needed:  Σ_k var_k = (Σ_i |d_i|² − Σ_k (Σ_i d_ik)² / N) / (N − 1) -> four wave sums
  wave_sum_n<4>(s);
  const double v = fmax(0.0, (s[3] - (s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) * inv) * den);
  return fast_sqrt(v);   // Manifolds.std: root of the corrected Fréchet variance (sum of the coordinate variances)
}
template <int PPL>
__device__ __forceinline__ double spread_r2(const double (&t)[PPL][2], const bool (&act)[PPL], double inv, double den) {
  const double x0 = readlane_f64(t[0][0], 0), y0 = readlane_f64(t[0][1], 0);
  double s[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const double dx = t[k][0] - x0, dy = t[k][1] - y0;
    if (act[k]) { s[0] += dx; s[1] += dx * dx; s[2] += dy; s[3] += dy * dy; }
  }
  wave_sum_n<4>(s);
  const double vx = fmax(0.0, (s[1] - s[0] * s[0] * inv) * den);
  const double vy = fmax(0.0, (s[3] - s[2] * s[2] * inv) * den);
  return fast_sqrt(vx + vy);
}

struct P2P2 {
  static constexpr int DF = 3, DT = 3, DZ = 3, NL = 6, NK = 9;
  static constexpr int kHypoDir = 2;   // multihypo over the SECOND pose of the factor: the fractional side follows the row's direction
                                       // (dir 0: the target is one of the candidates; dir 1: the fixed pose is drawn per particle)
  static constexpr bool kUniqueRoot = true;   // r(z; p, ·) = 0 has exactly one solution: the start point cannot reach the proposal
  struct Consts { double mu[3]; double L[6]; int dir; };
  __device__ static __forceinline__ Consts load(const ConvArgs& a, int f, int dr) {
    Consts K;
#pragma unroll
    for (int k = 0; k < 3; ++k) K.mu[k] = a.mu[3 * f + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) K.L[k] = a.L[6 * f + k];
    K.dir = dr;
    return K;
  }
  // the same constants from the block's LDS image [μ(3), L(6)] (k_conv_flat)
  __device__ static __forceinline__ Consts from_lds(const double* sk, int dr) {
    Consts K;
#pragma unroll
    for (int k = 0; k < 3; ++k) K.mu[k] = sk[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) K.L[k] = sk[3 + k];
    K.dir = dr;
    return K;
  }
  __device__ static __forceinline__ void measurement(const Consts& K, const double (&xi)[3], double (&z)[3]) {
    z[0] = K.mu[0] + K.L[0] * xi[0];
    z[1] = K.mu[1] + K.L[1] * xi[0] + K.L[2] * xi[1];
    z[2] = K.mu[2] + K.L[3] * xi[0] + K.L[4] * xi[1] + K.L[5] * xi[2];
  }
  __device__ static __forceinline__ void canonical(double (&t)[3]) { t[2] = wrap_pi(t[2]); }
  // do the inflation cycles (entropy + re-solve) apply?  Only to Nelder-Mead, whose answer (to its g_tol of 1e-8 on the simplex spread)
  // depends on where it starts.  CLOSED_FORM and NEWTON return the unique root directly; GAUSS_NEWTON iterates on the residual functor
  // from the belief point until max|r| <= tol (1e-12): the root is unique, so no start point -- jittered or not -- and no number of
  // cycles can move the converged answer by more than the solver tolerance, and the entropy / re-solve rounds are not run.
  __device__ static __forceinline__ bool needs_cycles(int solver, const Consts& K) {
    return solver == kSolverNelderMead && K.dir != kDirPrior;
  }
  struct Aux {};
  __device__ static __forceinline__ Aux init_aux(const double (&)[3]) { return Aux{}; }
  __device__ static __forceinline__ void finalize(double (&)[3], const Aux&) {}
  // tangent coordinates of a point about the belief's particle 0 (the spread statistic, chunk by chunk: k_conv_big)
  struct Ref { double c[3]; };
  __device__ static __forceinline__ Ref make_ref(const double (&t0)[3], const Aux&) { return Ref{{t0[0], t0[1], t0[2]}}; }
  __device__ static __forceinline__ void tangent(const Ref& r, const double (&t)[3], const Aux&, double (&d)[3]) {
    d[0] = t[0] - r.c[0]; d[1] = t[1] - r.c[1]; d[2] = wrap_pi(t[2] - r.c[2]);
  }
  template <int PPL>
  __device__ static __forceinline__ double spread(const double (&t)[PPL][3], const Aux (&)[PPL], const bool (&act)[PPL], double inv, double den) {
    return spread_se2<PPL>(t, act, inv, den);
  }
  __device__ static __forceinline__ void add_entropy(double (&t)[3], Aux&, double spread, const double (&u)[3]) {
    double s, c; fast_sincos(t[2], &s, &c);
    const double ex = spread * (u[0] - 0.5), ey = spread * (u[1] - 0.5), et = spread * (u[2] - 0.5);
    t[0] += c * ex - s * ey; t[1] += s * ex + c * ey; t[2] = wrap_pi(t[2] + et);
  }

  // The root of the residual  r(z; p, q) = ( p.t + R(θp) z_t - q.t , wrap(θp + zθ - θq) )  (SURVEY A.5), per particle, for both
  // directions in ONE branch-free form (the direction is a per-thread value in k_conv_flat, where a wave spans two table rows):
  //   dir 0 (solve q): a = p ∘ exp_ϵ(z) = (p.t + R(θp) z_t, θp + zθ)
  //   dir 1 (solve p): a = (q.t - R(θq - zθ) z_t, θq - zθ)
  //   prior row:       a = z   = dir 0 about the identity pose (R(0) z_t + 0 is exact)
  // Rounding is pinned by explicit fma (the packed sweep, the wave-per-row kernel and the per-factor entry points agree bit for bit).
  struct Prep { double a0, a1, a2; };
  __device__ static __forceinline__ Prep prepare(const Consts& K, const double (&z)[3], const double (&fxc)[3]) {
    const bool pr = K.dir == kDirPrior, back = K.dir == 1;
    const double f0 = pr ? 0.0 : fxc[0], f1 = pr ? 0.0 : fxc[1], f2 = pr ? 0.0 : fxc[2];
    const double sg = back ? -1.0 : 1.0;          // a = f ± (...) as one fma with an exact ±1 factor: the rounding of the sum / difference
    Prep P;
    P.a2 = __builtin_fma(sg, z[2], f2);
    const double thr = back ? P.a2 : f2;
    double s, c; fast_sincos(thr, &s, &c);
    const double vx = __builtin_fma(c, z[0], -(s * z[1])), vy = __builtin_fma(s, z[0], c * z[1]);
    P.a0 = __builtin_fma(sg, vx, f0);
    P.a1 = __builtin_fma(sg, vy, f1);
    return P;
  }
  // the residual FUNCTOR itself (src/factors/Pose2D.jl:51-67 / PriorPose2.jl:37-47, through points) at the target point t.
  // Fn = what does not change over the iterates of a root-find: the fixed point (or the prior's sample point) and sin/cos of z_θ
  struct Fn { Se2 F; double sz, cz; };
  __device__ static __forceinline__ Fn functor_setup(const Consts& K, const double (&z)[3], const double (&fxc)[3]) {
    Fn f;
    if (K.dir == kDirPrior) { f.F = se2_from_coords(z[0], z[1], z[2]); f.sz = 0.0; f.cz = 1.0; }
    else { f.F = se2_from_coords(fxc[0], fxc[1], fxc[2]); fast_sincos(z[2], &f.sz, &f.cz); }
    return f;
  }
  __device__ static __forceinline__ void functor(const Consts& K, const Fn& f, const double (&z)[3], const double (&t)[3], double (&r)[3],
                                                 double* st = nullptr, double* ct = nullptr) {
    const Se2 T = se2_from_coords(t[0], t[1], t[2]);
    if (st) { *st = T.s; *ct = T.c; }   // (the Gauss-Newton step of dir 1 needs R'(θ) at the same θ)
    if (K.dir == kDirPrior) residual_priorpose2(f.F, T, r);
    else if (K.dir == 0) residual_pose2pose2(z[0], z[1], f.cz, f.sz, f.F, T, r);
    else residual_pose2pose2(z[0], z[1], f.cz, f.sz, T, f.F, r);
  }
  // status of a directly returned root: max|r| of the functor there against tol
  __device__ static __forceinline__ int verify(const Consts& K, const double (&z)[3], const double (&fxc)[3], const double (&t)[3], const Aux&, double tol) {
    // ONE branch-free evaluation for the three row kinds (the packed sweep's waves span rows of both directions): the residual of
    // gauss_newton's predicted-pose form -- S = the pose the factor predicts for q (dir 1: from the returned p), G = q -- which is the
    // functor's residual up to the sign of both parts; |r_θ| <= tol is decided on the unit vector (U11, U21) itself when it is small
    const bool back = K.dir == 1, prior = K.dir == kDirPrior;
    const Fn f = functor_setup(K, z, fxc);
    const Se2 T = se2_from_coords(t[0], t[1], t[2]);
    const double zx = prior ? 0.0 : z[0], zy = prior ? 0.0 : z[1];
    const double Xx = back ? T.x : f.F.x, Xy = back ? T.y : f.F.y, Xc = back ? T.c : f.F.c, Xs = back ? T.s : f.F.s;
    const double Mx = Xx + Xc * zx - Xs * zy, My = Xy + Xs * zx + Xc * zy, Mc = Xc * f.cz - Xs * f.sz, Ms = Xs * f.cz + Xc * f.sz;
    const double Gx = back ? f.F.x : T.x, Gy = back ? f.F.y : T.y, Gc = back ? f.F.c : T.c, Gs = back ? f.F.s : T.s;
    const double U11 = Mc * Gc + Ms * Gs, U21 = Mc * Gs - Ms * Gc;
    const bool small = U11 > 0.0 && fabs(U21) < 1e-8;
    const double r2 = small ? U21 : fast_atan2(U21, U11);
    return fmax(fabs(Gx - Mx), fmax(fabs(Gy - My), fabs(r2))) <= tol ? 0 : 1;
  }
  // Gauss-Newton on the functor (the oracle's p2p2_newton): evaluate r at the current point, step on the group.
  // Round 6 (i): the iterate carries (cos θ, sin θ) -- the heading residual is atan2(U21, U11) of a UNIT vector (U11, U21) = (cos r_θ, sin r_θ),
  // so the heading update θ += r_θ is the rotation of (c, s) by (U11, U21): no sincos of the new iterate; the ANGLE r_θ (the accumulated
  // output heading) costs one atan2 on the first iterate, from the second on |r_θ| < 1e-8 and r_θ = U21 to 1e-24.
  // Round 6 (ii): ONE loop body for both directions (a wave of the packed sweep spans rows of both), as P3P3::gauss_newton: the iteration
  // lives in the PREDICTED pose of q -- dir 0 / prior: the state S is q itself, the target G = F ∘ exp(z); dir 1: S = p ∘ exp(z) of the
  // iterate p, G = the fixed q -- with r = (G.t − S.t, angle of R_Sᵀ R_G) (dir 1: the functor's residual with both signs flipped; the
  // test is on max|r|) and the exact group update S.t += r_t, R_S ← R_S R(r_θ).  The oracle's dir-1 step linearises the translation
  // (J13, J23) and needs a third evaluation whenever the heading moved; this one lands on the root from any start: two evaluations.
  __device__ static __forceinline__ int gauss_newton(const Consts& K, const double (&z)[3], const double (&fxc)[3], double (&t)[3], int max_iters, double tol) {
    const bool back = K.dir == 1, prior = K.dir == kDirPrior;
    const Fn f = functor_setup(K, z, fxc);                     // (prior row: F = the sample point, z's rotation the identity)
    const Se2 T = se2_from_coords(t[0], t[1], t[2]);
    const double zx = prior ? 0.0 : z[0], zy = prior ? 0.0 : z[1];
    // M = X ∘ exp(z), X = the fixed pose (dir 0: the target is predicted from it) or the start iterate (dir 1: the state is)
    const double Xx = back ? T.x : f.F.x, Xy = back ? T.y : f.F.y, Xc = back ? T.c : f.F.c, Xs = back ? T.s : f.F.s;
    const double Mx = Xx + Xc * zx - Xs * zy, My = Xy + Xs * zx + Xc * zy, Mc = Xc * f.cz - Xs * f.sz, Ms = Xs * f.cz + Xc * f.sz;
    double Sx = back ? Mx : T.x, Sy = back ? My : T.y, Sc = back ? Mc : T.c, Ss = back ? Ms : T.s;
    const double Gx = back ? f.F.x : Mx, Gy = back ? f.F.y : My, Gc = back ? f.F.c : Mc, Gs = back ? f.F.s : Ms;
    double ang = back ? t[2] + z[2] : t[2];                    // the state's heading as an angle (the output accumulates the steps)
    int st = 1;
    for (int it = 0; it < max_iters; ++it) {
      const double U11 = Sc * Gc + Ss * Gs, U21 = Sc * Gs - Ss * Gc;
      const double r0 = Gx - Sx, r1 = Gy - Sy;
      const bool small = U11 > 0.0 && fabs(U21) < 1e-8;
      const double r2 = small ? U21 : fast_atan2(U21, U11);
      if (fmax(fabs(r0), fmax(fabs(r1), fabs(r2))) <= tol) { st = 0; break; }
      const double c0 = Sc, s0 = Ss;
      Sx += r0; Sy += r1; ang += r2;
      Sc = c0 * U11 - s0 * U21; Ss = s0 * U11 + c0 * U21;       // rotation by +r_θ
    }
    // the iterate itself: dir 0 / prior S; dir 1  R_p = R_S R(z_θ)ᵀ, p.t = S.t − R_p z_t, θ_p = θ_S − z_θ
    const double Pc = Sc * f.cz + Ss * f.sz, Ps = Ss * f.cz - Sc * f.sz;
    t[0] = back ? Sx - (Pc * zx - Ps * zy) : Sx;
    t[1] = back ? Sy - (Ps * zx + Pc * zy) : Sy;
    t[2] = back ? ang - z[2] : ang;
    return st;
  }

  template <int SOLVER>
  __device__ static __forceinline__ int solve(const Consts& K, const Prep& P, const double (&z)[3], const double (&fxc)[3],
                                              double (&t)[3], Aux&, int max_iters, double tol) {
    int st = 0;
    if (K.dir == kDirPrior || SOLVER == kSolverClosedForm || SOLVER == kSolverNewton) {
      // PriorPose2 row: the sample exp_ϵ(hat(μ + Lξ)) itself is the proposal; relative rows: the unique root
      t[0] = P.a0; t[1] = P.a1; t[2] = wrap_pi(P.a2);
      return 0;
    }
    if constexpr (SOLVER == kSolverGaussNewton) st = gauss_newton(K, z, fxc, t, max_iters, tol);
    else {
      P2P2Cost cost{z[0], z[1], 0.0, 0.0, 0.0, K.dir};
      if (K.dir == 0) { cost.a0 = P.a0; cost.a1 = P.a1; cost.a2 = P.a2; }
      else { cost.a0 = fxc[0]; cost.a1 = fxc[1]; cost.a2 = P.a2; }
      st = nelder_mead<3>(cost, t, max_iters, tol);
    }
    t[2] = wrap_pi(t[2]);
    return st;
  }
};

// ---- Pose2Point2BearingRange; DIR 0: pose fixed -> landmark target, DIR 1: landmark fixed -> pose target
template <int DIR>
struct BRCost {
  double b, rho; double fx[3];
  __device__ __forceinline__ double operator()(const double (&x)[DIR == 0 ? 2 : 3]) const {
    double r[2];
    if constexpr (DIR == 0) {
      const Se2 P = se2_from_coords(fx[0], fx[1], fx[2]);
      residual_bearingrange(b, rho, P, x[0], x[1], r);
    } else {
      const Se2 P = se2_from_coords(x[0], x[1], x[2]);
      residual_bearingrange(b, rho, P, fx[0], fx[1], r);
    }
    return r[0] * r[0] + r[1] * r[1];
  }
};

template <int DIR>
struct BR {
  static constexpr int DF = DIR == 0 ? 3 : 2, DT = DIR == 0 ? 2 : 3, DZ = 2, NL = 2, NK = 4;
  static constexpr int kHypoDir = DIR;  // multihypo over the landmark slot: DIR 0 target is fractional, DIR 1 fixed is fractional
  static constexpr bool kUniqueRoot = DIR == 0;   // pose direction: 2 equations / 3 unknowns, a ring of roots around the landmark
  struct Consts { double mu[2]; double sg[2]; };
  __device__ static __forceinline__ Consts load(const ConvArgs& a, int f, int) {
    Consts K; K.mu[0] = a.mu[2 * f]; K.mu[1] = a.mu[2 * f + 1]; K.sg[0] = a.L[2 * f]; K.sg[1] = a.L[2 * f + 1];
    return K;
  }
  __device__ static __forceinline__ Consts from_lds(const double* sk, int) {
    Consts K; K.mu[0] = sk[0]; K.mu[1] = sk[1]; K.sg[0] = sk[2]; K.sg[1] = sk[3];
    return K;
  }
  __device__ static __forceinline__ void measurement(const Consts& K, const double (&xi)[2], double (&z)[2]) {
    // rand(bearing), rand(range)  (BearingRange2D.jl:23).  sg >= 0: Normal(mu, sg).  sg < 0: Uniform(mu - |sg|, mu + |sg|)
    // (test/TestPoseAndPoint2Constraints.jl:95 uses Uniform(-π, π) bearings): the standard normal ξ is mapped through
    // its CDF, u = ½ erfc(-ξ/√2).
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (K.sg[k] >= 0.0) z[k] = K.mu[k] + K.sg[k] * xi[k];
      else z[k] = K.mu[k] - K.sg[k] * (erfc(-xi[k] * 0.70710678118654752440) - 1.0);
    }
  }
  __device__ static __forceinline__ void canonical(double (&t)[DT]) { if constexpr (DT == 3) t[2] = wrap_pi(t[2]); }
  // landmark direction: unique root, only the start-dependent solvers cycle; pose direction: every solver starts from the belief point
  __device__ static __forceinline__ bool needs_cycles(int solver, const Consts&) {
    return DIR == 1 || solver == kSolverNelderMead;   // (DIR 0 under GAUSS_NEWTON: unique root, see P2P2::needs_cycles)
  }
  struct Aux {};
  __device__ static __forceinline__ Aux init_aux(const double (&)[DT]) { return Aux{}; }
  __device__ static __forceinline__ void finalize(double (&)[DT], const Aux&) {}
  struct Ref { double c[DT]; };
  __device__ static __forceinline__ Ref make_ref(const double (&t0)[DT], const Aux&) {
    Ref r;
#pragma unroll
    for (int k = 0; k < DT; ++k) r.c[k] = t0[k];
    return r;
  }
  __device__ static __forceinline__ void tangent(const Ref& r, const double (&t)[DT], const Aux&, double (&d)[DT]) {
    d[0] = t[0] - r.c[0]; d[1] = t[1] - r.c[1];
    if constexpr (DT == 3) d[2] = wrap_pi(t[2] - r.c[2]);
  }
  template <int PPL>
  __device__ static __forceinline__ double spread(const double (&t)[PPL][DT], const Aux (&)[PPL], const bool (&act)[PPL], double inv, double den) {
    if constexpr (DT == 3) return spread_se2<PPL>(t, act, inv, den);
    else return spread_r2<PPL>(t, act, inv, den);
  }
  __device__ static __forceinline__ void add_entropy(double (&t)[DT], Aux&, double spread, const double (&u)[DT]) {
    if constexpr (DT == 3) {
      double s, c; fast_sincos(t[2], &s, &c);
      const double ex = spread * (u[0] - 0.5), ey = spread * (u[1] - 0.5), et = spread * (u[2] - 0.5);
      t[0] += c * ex - s * ey; t[1] += s * ex + c * ey; t[2] = wrap_pi(t[2] + et);
    } else { t[0] += spread * (u[0] - 0.5); t[1] += spread * (u[1] - 0.5); }
  }
  // Solver form of the residual (src/factors/BearingRange2D.jl:48-64): pl = R(θp)ᵀ (l − p.t) has norm n = ‖l − p.t‖ and angle
  // ψ − θp with ψ = atan2(l − p.t) the world bearing, so  r = ( sym_rem(b − (ψ − θp)), ρ − n )  without a sin/cos per evaluation
  // (same function as residual_bearingrange up to rounding; the residual entry points and GAUSS_NEWTON keep the literal form).
  //   DIR 0 (landmark): unique root  l* = p.t + ρ (cos, sin)(θp + b), prepared once (one sincos) -- CLOSED_FORM and NEWTON return it.
  //   DIR 1 (pose):     2 equations / 3 unknowns.  The block step keeps the ray landmark -> pose: move along it to the measured
  //                     range, then turn to the measured bearing (what the minimum-norm Gauss-Newton step approaches for ρ >> 1).
  struct Prep { double a0, a1; };
  __device__ static __forceinline__ Prep prepare(const Consts&, const double (&z)[2], const double (&fx)[DF]) {
    Prep P; P.a0 = 0.0; P.a1 = 0.0;
    if constexpr (DIR == 0) {
      double s, c; fast_sincos(fx[2] + z[0], &s, &c);
      P.a0 = fx[0] + z[1] * c; P.a1 = fx[1] + z[1] * s;
    }
    return P;
  }
  // the residual FUNCTOR itself at the target point t (pose fixed / landmark target, or the reverse)
  __device__ static __forceinline__ void functor(const double (&z)[2], const double (&fx)[DF], const double (&t)[DT], double (&r)[2]) {
    if constexpr (DIR == 0) residual_bearingrange(z[0], z[1], se2_from_coords(fx[0], fx[1], fx[2]), t[0], t[1], r);
    else residual_bearingrange(z[0], z[1], se2_from_coords(t[0], t[1], t[2]), fx[0], fx[1], r);
  }
  __device__ static __forceinline__ int verify(const Consts&, const double (&z)[2], const double (&fx)[DF], const double (&t)[DT], const Aux&, double tol) {
    double r[2]; functor(z, fx, t, r);
    return fmax(fabs(r[0]), fabs(r[1])) <= tol ? 0 : 1;
  }
  // Gauss-Newton on the functor (the oracle's br_newton): r = (sym_rem(b - atan2(pl)), rho - |pl|), pl = R(theta_p)^T (l - p.t), at every iterate;
  //   DIR 0: exact Newton step in the pose-frame polar chart of the landmark, (phi, n) += (r0, r1);  DIR 1: the block step along the ray.
  // Round 6 (as P2P2 / P3P3): an iterate is evaluated in the form its step needs.  The RANGE residual rho - |pl| comes first (one reciprocal
  // square root, shared with the step); the BEARING residual is evaluated only where the test max|r| <= tol can pass (wave-uniform: a
  // jittered start is never within 1e-12 of the measured range), and then as the angle of pl rotated by -b, whose small-angle branch
  // (-w_y / w_x for |w_y| < 1e-8 w_x: the verification iterate) needs no atan2.  The frame of the next iterate is carried: DIR 1 -- the
  // heading after the block step is (world bearing of the ray) - b, its (cos, sin) the unit ray rotated by -b: no sincos of the new
  // heading; DIR 0 -- phi + r0 = b (mod 2 pi) and n + r1 = rho, so the step lands on rho (cos b, sin b) in the fixed pose's frame: no atan2.
  // One sincos of the bearing sample per call; per cycle of the pose direction one atan2 (the heading itself, which the spread statistic and
  // the output need) instead of three and no sincos instead of two (k_conv<BR<1>, 3>: profiles/r06_other_factors_trace.md).
  __device__ static __forceinline__ int gauss_newton(const double (&z)[2], const double (&fx)[DF], double (&t)[DT], int max_iters, double tol) {
    double sz, cz; fast_sincos(z[0], &sz, &cz);
    double s = 0.0, c = 1.0;                                   // the frame the residual is taken in: the fixed pose (DIR 0) / the iterate (DIR 1)
    if constexpr (DIR == 0) fast_sincos(fx[2], &s, &c);
    bool fresh = true;                                         // DIR 1: (c, s) of the iterate's heading not carried yet (the start point)
    for (int it = 0; it < max_iters; ++it) {
      // landmark - pose translation in the world frame, its squared norm and 1 / norm
      const double dx = DIR == 0 ? t[0] - fx[0] : fx[0] - t[0], dy = DIR == 0 ? t[1] - fx[1] : fx[1] - t[1];
      const double n2 = dx * dx + dy * dy;
      double y = __builtin_amdgcn_rsq(n2);
      y = y * __builtin_fma(-0.5 * n2 * y, y, 1.5);
      y = y * __builtin_fma(-0.5 * n2 * y, y, 1.5);
      const bool ok = n2 > 0.0;
      const double r1 = z[1] - (ok ? n2 * y : 0.0);
      if (__builtin_amdgcn_ballot_w64(fabs(r1) <= tol) != 0) {   // somebody may be at a root: the bearing residual
        if constexpr (DIR == 1) { if (fresh) fast_sincos(t[2], &s, &c); }
        const double plx = c * dx + s * dy, ply = c * dy - s * dx;
        const double wx = plx * cz + ply * sz, wy = ply * cz - plx * sz;     // pl rotated by -b: its angle is -(b - atan2(pl))
        const bool small = wx > 0.0 && fabs(wy) < 1e-8 * wx;
        double r0;
        if (__builtin_amdgcn_ballot_w64(!small) == 0) r0 = -wy * fast_rcp(wx);
        else r0 = small ? -wy * fast_rcp(wx) : sym_rem(z[0] - fast_atan2(ply, plx));
        if (fmax(fabs(r0), fabs(r1)) <= tol) return 0;
      }
      if constexpr (DIR == 0) {       // (phi, n) += (r0, r1) = (b, rho) in the pose frame
        const double qx = z[1] * cz, qy = z[1] * sz;
        t[0] = fx[0] + c * qx - s * qy; t[1] = fx[1] + s * qx + c * qy;
      } else {                        // the block step along the ray (ring_step), the new heading's (cos, sin) = the unit ray rotated by -b
        const double k = ok ? z[1] * y : 0.0;
        t[0] = ok ? fx[0] - k * dx : fx[0] - z[1]; t[1] = fx[1] - k * dy;
        if constexpr (DT == 3) t[2] = (ok ? fast_atan2(dy, dx) : 0.0) - z[0];
        const double ux = ok ? dx * y : 1.0, uy = ok ? dy * y : 0.0;
        c = ux * cz + uy * sz; s = uy * cz - ux * sz;
        fresh = false;
      }
    }
    return 1;
  }
  // pose direction: move along the ray landmark -> pose to the measured range, then turn to the measured bearing.  One reciprocal
  // square root (v_rsq_f64 + two Newton steps) serves the unit vector; the world bearing is atan2 of the ray itself.
  __device__ static __forceinline__ void ring_step(const double (&z)[2], const double (&fx)[DF], double (&t)[DT]) {
    const double dx = fx[0] - t[0], dy = fx[1] - t[1];
    const double n2 = dx * dx + dy * dy;
    double y = __builtin_amdgcn_rsq(n2);
    y = y * __builtin_fma(-0.5 * n2 * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * n2 * y, y, 1.5);
    const bool ok = n2 > 0.0;                                   // pose on the landmark: leave along +x
    const double k = ok ? z[1] * y : 0.0;
    t[0] = ok ? fx[0] - k * dx : fx[0] - z[1]; t[1] = fx[1] - k * dy;
    if constexpr (DT == 3) t[2] = (ok ? fast_atan2(dy, dx) : 0.0) - z[0];
  }
  template <int SOLVER>
  __device__ static __forceinline__ int solve(const Consts&, const Prep& P, const double (&z)[2], const double (&fx)[DF],
                                              double (&t)[DT], Aux&, int max_iters, double tol) {
    int st = 0;
    if constexpr (DIR == 0 && (SOLVER == kSolverClosedForm || SOLVER == kSolverNewton)) { t[0] = P.a0; t[1] = P.a1; return 0; }
    else if constexpr (SOLVER == kSolverClosedForm) {
      ring_step(z, fx, t);
    } else if constexpr (SOLVER == kSolverNewton) {
      // pose direction: the block step from ANY start lands exactly on the member of the ring of roots that the start selects (the
      // closed form above IS that step); the residual there is evaluated only for the status array (verify_ring, after the last cycle)
      ring_step(z, fx, t);
    } else if constexpr (SOLVER == kSolverGaussNewton) {
      st = gauss_newton(z, fx, t, max_iters, tol);
    } else {
      BRCost<DIR> cost{z[0], z[1], {fx[0], fx[1], DF == 3 ? fx[DF - 1] : 0.0}};
      st = nelder_mead<DT>(cost, t, max_iters, tol);
    }
    if constexpr (DT == 3) t[2] = wrap_pi(t[2]);
    return st;
  }
};

// ---- Pose3Pose3.  Belief blocks hold coordinates (t, ω); inside the kernel the rotation of every particle lives as a
// unit quaternion (Aux) from load to store, so the inflation cycles never go through Exp/Log round trips, and the root
// (a, qa) of the residual  r = ( p.t + R_p z_t − q.t , Log(R_qᵀ R_p Exp(z_ω)) )  is prepared once per particle:
//   dir 0 (solve q): qa = q_p ⊗ q_z,        a = p.t + R_p z_t   (the root itself)
//   dir 1 (solve p): qa = q_q ⊗ conj(q_z),  a = q.t             (root translation = a − R(qa) z_t)
// The Newton rotation residual is conj(q_T) ⊗ qa: the same angle as the reference's Log(R_qᵀ R_p Z) (for dir 1 the
// vector is that residual rotated by Z, which changes neither its norm nor the root).
// Nelder-Mead mode evaluates Σr² through 3x3 frames, the residual exactly as src/factors/Pose3Pose3.jl:17-29 composes it
// (measured: 191 ms per helix sweep against 264 ms for a quaternion cost, whose inverse-trig call raises the register
// pressure of the 32 inlined evaluations; the Newton / closed-form modes never evaluate a Log).
struct P3P3Cost {
  double zt[3]; double Z[9]; Se3 F; int dir;
  __device__ __forceinline__ double operator()(const double (&x)[6]) const {
    Se3 T; se3_from_coords(x, T);
    double r[6];
    if (dir == 0) residual_pose3pose3(zt, Z, F, T, r); else residual_pose3pose3(zt, Z, T, F, r);
    double s = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) s += r[k] * r[k];
    return s;
  }
};

struct P3P3 {
  static constexpr int DF = 6, DT = 6, DZ = 6, NL = 21, NK = 27;
  static constexpr int kHypoDir = -1;
  static constexpr bool kUniqueRoot = true;
  struct Consts { double mu[6]; const double* L; int dir; };
  __device__ static __forceinline__ Consts load(const ConvArgs& a, int f, int dr) {
    Consts K;
#pragma unroll
    for (int k = 0; k < 6; ++k) K.mu[k] = a.mu[6 * f + k];
    K.L = a.L + 21 * (size_t)f;  // 21 wave-uniform doubles, read through the scalar cache at use
    K.dir = dr;
    return K;
  }
  __device__ static __forceinline__ Consts from_lds(const double* sk, int dr) {
    Consts K;
#pragma unroll
    for (int k = 0; k < 6; ++k) K.mu[k] = sk[k];
    K.L = sk + 6;
    K.dir = dr;
    return K;
  }
  __device__ static __forceinline__ void measurement(const Consts& K, const double (&xi)[6], double (&z)[6]) {
    int p = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double s = K.mu[k];
#pragma unroll
      for (int j = 0; j <= k; ++j) s += K.L[p++] * xi[j];
      z[k] = s;
    }
  }
  __device__ static __forceinline__ void canonical(double (&)[6]) {}   // finalize() writes the principal rotation vector
  __device__ static __forceinline__ bool needs_cycles(int solver, const Consts& K) {
    return solver == kSolverNelderMead && K.dir != kDirPrior;
  }
  struct Aux { double q[4]; };
  // start point u0 -> state.  The reference takes X0c = vee(log(ϵ, u0)) of the start point, and Manifolds' log returns θ = π exactly
  // for rotations with cos θ + 1 <= √eps: the same snap is applied to the quaternion (w = 0, unit vector part).
  __device__ static __forceinline__ Aux init_aux(const double (&t)[6]) {
    Aux A; quat_exp(&t[3], A.q);
    if (2.0 * A.q[0] * A.q[0] <= kSqrtEps) {
      const double inv = 1.0 / fast_sqrt(A.q[1] * A.q[1] + A.q[2] * A.q[2] + A.q[3] * A.q[3]);
      A.q[0] = 0.0; A.q[1] *= inv; A.q[2] *= inv; A.q[3] *= inv;
    }
    return A;
  }
  __device__ static __forceinline__ void finalize(double (&t)[6], const Aux& A) { quat_log(A.q, &t[3]); }
  struct Ref { double c[3]; double q[4]; };
  __device__ static __forceinline__ Ref make_ref(const double (&t0)[6], const Aux& A0) {
    return Ref{{t0[0], t0[1], t0[2]}, {A0.q[0], A0.q[1], A0.q[2], A0.q[3]}};
  }
  __device__ static __forceinline__ void tangent(const Ref& r, const double (&t)[6], const Aux& A, double (&d)[6]) {
    double e[4];
    quat_cmul(r.q, A.q, e); quat_log(e, d + 3);
    d[0] = t[0] - r.c[0]; d[1] = t[1] - r.c[1]; d[2] = t[2] - r.c[2];
  }

  // std of the tangent coordinates about particle 0: translation differences and Log(R0ᵀ R_i)
  template <int PPL>
  __device__ static __forceinline__ double spread(const double (&t)[PPL][6], const Aux (&A)[PPL], const bool (&act)[PPL], double inv, double den) {
    double c0[3], q0[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) c0[k] = readlane_f64(t[0][k], 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) q0[k] = readlane_f64(A[0].q[k], 0);
    double s[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = 0.0;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      double e[4], d[6];
      quat_cmul(q0, A[k].q, e); quat_log(e, d + 3);
      d[0] = t[k][0] - c0[0]; d[1] = t[k][1] - c0[1]; d[2] = t[k][2] - c0[2];
      if (act[k]) {
#pragma unroll
        for (int j = 0; j < 6; ++j) { s[2 * j] += d[j]; s[2 * j + 1] += d[j] * d[j]; }
      }
    }
    wave_sum_n<12>(s);
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += fmax(0.0, (s[2 * j + 1] - s[2 * j] * s[2 * j] * inv) * den);
    return fast_sqrt(acc);
  }
  // The root (a, qa) of the residual, prepared once per particle for both directions (cf. P2P2::Prep): with the rotation
  // solved first, the translation residual is affine with R at the root rotation, so R(qa) z_t is loop-invariant:
  //   dir 0 (solve q): qa = q_p ⊗ q_z,        a = p.t + R_p z_t
  //   dir 1 (solve p): qa = q_q ⊗ conj(q_z),  a = q.t − R(qa) z_t
  //   prior row:       qa = Exp(z_ω),          a = z_t
  struct Prep { double a[3], qa[4]; };
  __device__ static __forceinline__ Prep prepare(const Consts& K, const double (&z)[6], const double (&fxc)[6]) {
    Prep P;
    double qz[4];
    quat_exp(&z[3], qz);
    if (K.dir == kDirPrior) {  // PriorPose3 row: the sample point exp_ϵ(hat z)
#pragma unroll
      for (int k = 0; k < 3; ++k) P.a[k] = z[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) P.qa[k] = qz[k];
      return P;
    }
    // both directions in ONE branch-free form (a wave of the packed sweep spans rows of both; cf. P2P2::prepare):
    //   qa = q_F (x) (dir 1 ? conj(q_z) : q_z),   a = F.t +- R(dir 1 ? qa : q_F) z_t
    double qF[4], v[3], qs[4], qr[4];
    quat_exp(&fxc[3], qF);
    const bool back = K.dir != 0;
    const double sg = back ? -1.0 : 1.0;
    qs[0] = qz[0]; qs[1] = sg * qz[1]; qs[2] = sg * qz[2]; qs[3] = sg * qz[3];
    quat_mul(qF, qs, P.qa);
#pragma unroll
    for (int k = 0; k < 4; ++k) qr[k] = back ? P.qa[k] : qF[k];
    quat_rot(qr, z, v);
#pragma unroll
    for (int k = 0; k < 3; ++k) P.a[k] = __builtin_fma(sg, v[k], fxc[k]);
    return P;
  }
  // u0 ∘ exp_ϵ(hat e), e = spread·(u − ½):  t += R e_t,  R ← R Exp(e_ω)
  __device__ static __forceinline__ void add_entropy(double (&t)[6], Aux& A, double spread, const double (&u)[6]) {
    double e[6], v[3], qe[4], qn[4];
#pragma unroll
    for (int k = 0; k < 6; ++k) e[k] = spread * (u[k] - 0.5);
    quat_rot(A.q, e, v);
    t[0] += v[0]; t[1] += v[1]; t[2] += v[2];
    quat_exp(e + 3, qe); quat_mul(A.q, qe, qn);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.q[k] = qn[k];
  }
  // the residual FUNCTOR itself (src/factors/Pose3Pose3.jl:17-29 / Pose3D.jl:15-19, through 3x3 frames) at the target (t, R(q))
  __device__ static __forceinline__ void quat_to_mat(const double (&q)[4], double* R) {   // column-major
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - 2.0 * (y * y + z * z); R[1] = 2.0 * (x * y + w * z);       R[2] = 2.0 * (x * z - w * y);
    R[3] = 2.0 * (x * y - w * z);       R[4] = 1.0 - 2.0 * (x * x + z * z); R[5] = 2.0 * (y * z + w * x);
    R[6] = 2.0 * (x * z + w * y);       R[7] = 2.0 * (y * z - w * x);       R[8] = 1.0 - 2.0 * (x * x + y * y);
  }
  __device__ static __forceinline__ void functor(const Consts& K, const double (&z)[6], const double* Z, const Se3& F, const Se3& T, double (&r)[6]) {
    if (K.dir == kDirPrior) { Se3 M; se3_from_coords(z, M); residual_priorpose3(M, T, r); }
    else if (K.dir == 0) residual_pose3pose3(z, Z, F, T, r);
    else residual_pose3pose3(z, Z, T, F, r);
  }
  __device__ static __forceinline__ int verify(const Consts& K, const double (&z)[6], const double (&fxc)[6], const double (&t)[6], const Aux& A, double tol) {
    // ONE branch-free evaluation for the three row kinds, on unit quaternions (the predicted-pose residual of gauss_newton below: the
    // functor's residual up to the sign of both parts; round 6 -- the 3x3 functor evaluated under three divergent branches cost 44 us of
    // the 92 us of a helix sweep with a status array): S = the pose the factor predicts for q (dir 1: from the returned p), G = q
    const bool back = K.dir == 1, prior = K.dir == kDirPrior;
    double qz[4], qF[4], Ft[3];
    quat_exp(&z[3], qz);
    quat_exp(&fxc[3], qF);
#pragma unroll
    for (int k = 0; k < 4; ++k) qF[k] = prior ? (k == 0 ? 1.0 : 0.0) : qF[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Ft[k] = prior ? 0.0 : fxc[k];
    double X[4], M[4], v[3], G[4], e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { X[k] = back ? A.q[k] : qF[k]; G[k] = back ? qF[k] : A.q[k]; }
    quat_mul(X, qz, M); quat_rot(X, z, v);
    quat_cmul(M, G, e);
    double mt = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) mt = fmax(mt, fabs((back ? Ft[k] : t[k]) - ((back ? t[k] : Ft[k]) + v[k])));
    const double n2e = e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
    if (mt > tol || 4.0 * n2e > 3.0 * tol * tol) return 1;    // |r_w|_inf >= 2 |vec e| / sqrt 3: not converged whatever the Log is
    double rw[3];
    if (n2e > 1e-16) quat_log(e, rw);                           // (only with a tolerance above 1e-8)
    else { const double k2 = 2.0 * fast_rcp(e[0]); rw[0] = k2 * e[1]; rw[1] = k2 * e[2]; rw[2] = k2 * e[3]; }
    return fmax(mt, fmax(fabs(rw[0]), fmax(fabs(rw[1]), fabs(rw[2])))) <= tol ? 0 : 1;
  }
  // Gauss-Newton on the functor (the oracle's p3p3_newton_pt): right-perturbation updates on the group that zero the residual,
  //   dir 0: R_q <- R_q Exp(r_w), q.t += r_t;   dir 1: R_p <- R_p Exp(-Z r_w), p.t <- q.t - R_p z_t
  // Round 6: the residual is evaluated ON UNIT QUATERNIONS -- r_w = Log(conj(q_q) (x) q_p (x) q_z), the same rotation as the functor's
  // Log(R_q^T R_p Exp(z_w)) (src/factors/Pose3Pose3.jl:17-29; the 3x3 form stays in `functor` / `verify` and the residual entry points),
  // r_t = p.t + R(q_p) z_t - q.t -- and the update is applied with the residual ROTATION e itself instead of Exp(Log(e)):
  //   dir 0: q_q <- q_q (x) e;   dir 1: Exp(-Z r_w) = q_z (x) conj(e) (x) conj(q_z), so q_p <- (q_p (x) q_z) (x) conj(e) (x) conj(q_z)
  // (the P2P2 iteration carries (cos, sin) the same way).  What an iterate costs: two or three quaternion products and one Log -- whose
  // inverse-trigonometric branch is skipped when every lane of the wave is at |vec e| < 1e-8 (Log e = 2 vec e / e_w to 1e-24: the
  // verification iterate) -- instead of a 3x3 frame from the quaternion, two 3x3 products, the matrix Log and an Exp per iterate
  // (the packed sweep on the 10k helix: 154.5 -> 102 us with this alone; profiles/r06_p3p3_gn.md).
  __device__ static __forceinline__ void quat_log_iter(const double (&q)[4], double* w) {
    const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (__builtin_amdgcn_ballot_w64(n2 > 1e-16) == 0) {   // wave-uniform: every active lane is at the root already
      const double k = 2.0 * fast_rcp(q[0]);
      w[0] = k * q[1]; w[1] = k * q[2]; w[2] = k * q[3];
      return;
    }
    quat_log(q, w);
  }
  // ONE loop body for both directions (a wave of the packed sweep spans rows of both: two branches would run one after the other).  The
  // iteration lives in the PREDICTED pose of q: (s, u) with target (Ts, Tu),
  //   dir 0 / prior:  (s, u) = (q.t, q_q) itself,                         target = F o exp(z) = (F.t + R_F z_t, q_F (x) q_z)
  //   dir 1:          (s, u) = (p.t + R_p z_t, q_p (x) q_z) of the iterate p,  target = the fixed q = (F.t, q_F)
  // residual e = conj(u) (x) Tu, r_t = Ts - s (dir 1: the functor's residual up to the sign of both parts -- the test is on max|r|);
  // update u <- u (x) e, s <- s + r_t.  In dir 1 this IS R_p <- R_p Exp(-Z r_w), p.t <- q.t - R_p z_t: u (x) e (x) conj(q_z) = q_p (x) q_z
  // (x) conj(e') (x) conj(q_z) with e' the functor's rotation.  The iterate p is recovered from (s, u) once, after the loop.
  __device__ static __forceinline__ int gauss_newton(const Consts& K, const double (&z)[6], const double (&fxc)[6], double (&t)[6], Aux& A, int max_iters, double tol) {
    const bool back = K.dir == 1, prior = K.dir == kDirPrior;
    double qz[4], qF[4], Ft[3];
    quat_exp(&z[3], qz);
    quat_exp(&fxc[3], qF);
#pragma unroll
    for (int k = 0; k < 4; ++k) qF[k] = prior ? (k == 0 ? 1.0 : 0.0) : qF[k];   // (prior row: the identity pose)
#pragma unroll
    for (int k = 0; k < 3; ++k) Ft[k] = prior ? 0.0 : fxc[k];
    // M = X (x) q_z, v = R(X) z_t with X = the fixed rotation (dir 0: the target's) or the start iterate's (dir 1: the state's)
    double X[4], M[4], v[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) X[k] = back ? A.q[k] : qF[k];
    quat_mul(X, qz, M); quat_rot(X, z, v);
    double u[4], Tu[4], s[3], Ts[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) { u[k] = back ? M[k] : A.q[k]; Tu[k] = back ? qF[k] : M[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) { s[k] = back ? t[k] + v[k] : t[k]; Ts[k] = back ? Ft[k] : Ft[k] + v[k]; }
    int st = 1;
    for (int it = 0; it < max_iters; ++it) {
      double e[4], r[6];
      quat_cmul(u, Tu, e);
      r[0] = Ts[0] - s[0]; r[1] = Ts[1] - s[1]; r[2] = Ts[2] - s[2];
      // The coordinates Log(e) are needed only where the test max|r| <= tol can pass: |r_w|_inf >= theta / sqrt 3 >= 2 |vec e| / sqrt 3, so
      // a lane with 4 |vec e|^2 > 3 tol^2 (or a translation residual above tol) is NOT converged whatever its Log is -- the same
      // decision without the inverse-trigonometric evaluation.  Wave-uniform: the start iterate skips the Log, the verification
      // iterate takes its small-angle branch (quat_log_iter); the full Log runs only for a wave with a lane in between.
      const double mt = fmax(fabs(r[0]), fmax(fabs(r[1]), fabs(r[2])));
      const double n2e = e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
      const bool undecided = !(mt > tol) && !(4.0 * n2e > 3.0 * tol * tol);
      if (__builtin_amdgcn_ballot_w64(undecided) != 0) {
        quat_log_iter(e, r + 3);
        if (fmax(mt, fmax(fabs(r[3]), fmax(fabs(r[4]), fabs(r[5])))) <= tol) { st = 0; break; }
      }
      double qn[4];
      quat_mul(u, e, qn);
      s[0] += r[0]; s[1] += r[1]; s[2] += r[2];
      // (renormalised: a product of unit quaternions drifts by an ulp per step; |q|^2 = 1 + eps, 1/|q| = 3/2 - |q|^2/2 to O(eps^2))
      const double nn = __builtin_fma(-0.5, qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3], 1.5);
#pragma unroll
      for (int k = 0; k < 4; ++k) u[k] = qn[k] * nn;
    }
    // the iterate itself: dir 0 / prior (s, u); dir 1  q_p = u (x) conj(q_z), p.t = s - R(q_p) z_t
    double qp[4], w[3];
    quat_mulc(u, qz, qp); quat_rot(qp, z, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) A.q[k] = back ? qp[k] : u[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = back ? s[k] - w[k] : s[k];
    return st;
  }
  template <int SOLVER>
  __device__ static __forceinline__ int solve(const Consts& K, const Prep& P, const double (&z)[6], const double (&fxc)[6],
                                              double (&t)[6], Aux& A, int max_iters, double tol) {
    int st = 0;
    if (K.dir == kDirPrior || SOLVER == kSolverClosedForm || SOLVER == kSolverNewton) {   // the prepared root itself
#pragma unroll
      for (int k = 0; k < 3; ++k) t[k] = P.a[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) A.q[k] = P.qa[k];
      return 0;
    }
    if constexpr (SOLVER == kSolverGaussNewton) st = gauss_newton(K, z, fxc, t, A, max_iters, tol);
    else if constexpr (SOLVER == kSolverNelderMead) {
      P3P3Cost cost;
#pragma unroll
      for (int k = 0; k < 3; ++k) cost.zt[k] = z[k];
      so3_exp(&z[3], cost.Z);
      se3_from_coords(fxc, cost.F);
      cost.dir = K.dir;
      quat_log(A.q, &t[3]);   // X0c = vee(log(ϵ,u0)): Nelder-Mead works on the (t, ω) coordinates
      st = nelder_mead<6>(cost, t, max_iters, tol);
      quat_exp(&t[3], A.q);
    }
    return st;
  }
};

// ------------------------------------------------------------------------------------------
// the convolution kernels
// ------------------------------------------------------------------------------------------
#ifndef ROME_P3_MINBLK
#define ROME_P3_MINBLK 1
#endif
#ifndef ROME_MIN_WAVES
#define ROME_MIN_WAVES 1
#endif
#ifndef ROME_NM_MINWAVES
#define ROME_NM_MINWAVES 4
#endif
#ifndef ROME_WPB
#define ROME_WPB 4   // wavefronts (= convolutions) per workgroup of k_conv
#endif

// slot k of lane `lane` -> particle: a lane owns NEIGHBOURING particles (2l, 2l+1), then (128 + 2l, 128 + 2l + 1), ...: the two
// particles that share a Box-Muller pair (rng_normals<3>) sit in one lane and are adjacent in every SoA row
template <int PPL>
__device__ __forceinline__ int slot_particle(int lane, int k) {
  if constexpr (PPL == 1) return lane;
  else return ((k >> 1) << 7) + 2 * lane + (k & 1);
}
// streaming (non-temporal) stores for data the launch does not read again
__device__ __forceinline__ void store_stream(double* p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_stream2(double* p, const double2& v) {
  typedef double dvec2 __attribute__((ext_vector_type(2)));
  const dvec2 vv = {v.x, v.y};
  __builtin_nontemporal_store(vv, reinterpret_cast<dvec2*>(p));
}

// separator rows are duplicated into the exchange buffer: block m of mirror_out for row c with mirror_map[c] = m >= 0
// (any number of rows), or -- the older form -- for the up to four rows listed in mirror_row
__device__ __forceinline__ int mirror_slot(const ConvArgs& a, int c_raw) {
  if (a.mirror_map) return a.mirror_map[c_raw];
  int m = -1;
  for (int q = 0; q < a.n_mirror; ++q) m = a.mirror_row[q] == c_raw ? q : m;
  return m;
}

template <class FP, int SOLVER, int PPL, bool LEAN> __device__ __forceinline__ void conv_wave_body(const ConvArgs& a, int blk);

// LEAN: the plain sweep -- in-kernel noise, all four table columns present, no multihypo / nullhypo rows.  The same code with
// those features compiled out: the table row is one 16-byte scalar load, nothing stands between the belief loads and the
// Philox / Box-Muller block, and the register allocation is not pinned by the feature paths.
template <class FP, int SOLVER, int PPL, bool LEAN>
// Nelder-Mead on the 2-D/3-D factors is latency-bound (long dependent select/compare chains): asking for 4 waves/SIMD
// (<= 128 VGPRs) is 5 % faster there; (the SE(3) kernels need their 256 VGPRs: capped at 3-4 waves/SIMD they spill and run 2.7x slower)
__global__ void __launch_bounds__(64 * ROME_WPB, (SOLVER == kSolverNelderMead && FP::DT <= 3) ? ROME_NM_MINWAVES : ((SOLVER != kSolverNelderMead && FP::DT == 6) ? ROME_P3_MINBLK : ROME_MIN_WAVES))
k_conv(const ConvArgs a) {
  conv_wave_body<FP, SOLVER, PPL, LEAN>(a, xcd_contiguous_block(blockIdx.x, gridDim.x));
}
template <class FP, int SOLVER, int PPL, bool LEAN>
__device__ __forceinline__ void conv_wave_body(const ConvArgs& a, int blk) {
  const int lane = threadIdx.x & 63;
  // no early exit: the (at most ROME_WPB - 1) surplus waves of the last block redo the last row and skip its stores, so that
  // no kernel-argument load has to wait for the n_conv comparison (all of them are issued together)
  const int c_raw = __builtin_amdgcn_readfirstlane(blk * ROME_WPB + (int)(threadIdx.x >> 6));
  const bool valid = c_raw < a.n_conv;
  const int c = valid ? c_raw : a.n_conv - 1;
  const int N = a.N;
  int f, dr, fv, tv;
  if (LEAN || a.rows4) {   // one 16-byte scalar load for the whole row
    const int4 row = *reinterpret_cast<const int4*>(a.rows4 + 4 * (size_t)c);
    f = row.x; fv = row.z; tv = row.w;
    dr = (FP::kHypoDir < 0 || FP::kHypoDir == 2) ? row.y : a.dir_all;   // bearing-range: the direction is the kernel's template argument
  } else {
    f = a.factor ? a.factor[c] : c;
    dr = a.dir ? a.dir[c] : a.dir_all;
    fv = a.fixed_var ? a.fixed_var[c] : c;
    tv = a.target_var ? a.target_var[c] : c;
  }
  const typename FP::Consts K = FP::load(a, f, dr);
  const double* __restrict__ fb = a.bel_fixed + (size_t)fv * FP::DF * N;
  const double* __restrict__ tb = a.bel_target + (size_t)tv * FP::DT * N;
  double* __restrict__ ob = a.out + (size_t)c * FP::DT * N;
  const uint64_t stream = a.stream_offset + (uint64_t)((!LEAN && a.row_stream) ? a.row_stream[c] : c);
  [[maybe_unused]] const int meas_blk = (!LEAN && a.meas_block) ? a.meas_block[c] : -1;

  double fx[PPL][FP::DF], t[PPL][FP::DT], z[PPL][FP::DZ];
  typename FP::Prep prep[PPL];
  typename FP::Aux aux[PPL];   // state a policy keeps beside the coordinates (Pose3: the rotation as a unit quaternion)
  bool act[PPL];
  [[maybe_unused]] double xi_odd[FP::DZ];   // normals of the odd slot, produced together with the even slot's (shared Philox calls)
  // measurement samples first (they depend on nothing but the convolution id), then the belief loads: the loaded particles
  // are then not live across the Philox / Box-Muller block (fewer registers at the kernel's pressure peak)
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int i = slot_particle<PPL>(lane, k);
    act[k] = i < N;
    const int ii = act[k] ? i : 0;  // idle lanes shadow particle 0 (keeps the math finite, never stored)
    double xi[FP::DZ];
    if (!LEAN && meas_blk >= 0) {   // the row's measurement samples live in a belief block (a message of a child clique)
      const double* nb = a.meas_base + (size_t)meas_blk * FP::DZ * N;
#pragma unroll
      for (int d = 0; d < FP::DZ; ++d) xi[d] = nb[d * N + ii];
    } else if (!LEAN && a.noise) {
      const double* nb = a.noise + (size_t)c * FP::DZ * N;
#pragma unroll
      for (int d = 0; d < FP::DZ; ++d) xi[d] = nb[d * N + ii];
    } else if constexpr (PPL >= 2) {
      // slots k (even) and k+1 of a lane are the neighbours 2j, 2j+1: they draw from the same Philox calls (rng_normals_pair)
      if ((k & 1) == 0) rng_normals_pair<FP::DZ>(a.seed, stream, (uint32_t)i, xi, xi_odd);
      else {
#pragma unroll
        for (int d = 0; d < FP::DZ; ++d) xi[d] = xi_odd[d];
      }
    } else {
      rng_normals<FP::DZ>(a.seed, stream, (uint32_t)ii, xi);
    }
    if (!LEAN && ((a.noise && a.noise_is_meas) || meas_blk >= 0)) {   // the caller sampled the measurement model itself (any SamplableBelief)
#pragma unroll
      for (int d = 0; d < FP::DZ; ++d) z[k][d] = xi[d];
    } else FP::measurement(K, xi, z[k]);
  }
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int i = slot_particle<PPL>(lane, k);
    const int ii = act[k] ? i : 0;
#pragma unroll
    for (int d = 0; d < FP::DF; ++d) fx[k][d] = fb[d * N + ii];
#pragma unroll
    for (int d = 0; d < FP::DT; ++d) t[k][d] = tb[d * N + ii];
    FP::canonical(t[k]);
    aux[k] = FP::init_aux(t[k]);
    prep[k] = FP::prepare(K, z[k], fx[k]);
  }

  int st[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) st[k] = 0;

  // ---- multihypo (IIF `multihypo=[1, w, 1-w]` on the second variable of a factor -- the landmark slot of a bearing-range
  //      factor, the second pose of a Pose2Pose2; ⚠IIF computeAcrossHypothesis!): per particle a categorical draw decides which
  //      candidate the measurement belongs to.
  //      side 1 (solve the first variable): the fixed particle comes from the drawn candidate.
  //      side 0 (solve this candidate): particles of the other hypothesis are not constrained by the factor: they keep
  //      their value and only receive entropy  spreadNH · ‖mean(this) - mean(other)‖ · (U-½)  (applied after the cycles).
  bool sel[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) sel[k] = true;
  double nh_spread = 0.0;
  [[maybe_unused]] int hd = FP::kHypoDir;   // which side of the factor is fractional (Pose2Pose2: by the row's direction)
  if constexpr (FP::kHypoDir >= 0 && !LEAN) {
    if constexpr (FP::kHypoDir == 2) hd = dr == 1 ? 1 : 0;
    const int av = (a.alt_var && dr != 2) ? a.alt_var[c] : -1;
    if (av >= 0) {  // wave-uniform
      const double w = a.hypo_w[c];
      const double* __restrict__ ab = (hd == 1 ? a.bel_fixed + (size_t)av * FP::DF * N : a.bel_target + (size_t)av * FP::DT * N);
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const int i = slot_particle<PPL>(lane, k), ii = act[k] ? i : 0;
        const u32x4 hw = philox4x32_10(u32x4{(uint32_t)ii, (uint32_t)stream, (uint32_t)(stream >> 32), (4u << 16)},
                                       (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        const bool primary = ((double)hw.x + 0.5) * (1.0 / 4294967296.0) < w;
        if (hd == 1) {
          if (!primary) {
#pragma unroll
            for (int d = 0; d < FP::DF; ++d) fx[k][d] = ab[d * N + ii];
            prep[k] = FP::prepare(K, z[k], fx[k]);
          }
        } else sel[k] = primary;
      }
      if (hd == 0) {
        double sm[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < PPL; ++k) if (act[k]) {
          const int i = slot_particle<PPL>(lane, k);
          sm[0] += t[k][0]; sm[1] += t[k][1]; sm[2] += ab[i]; sm[3] += ab[N + i];
        }
        wave_sum_n<4>(sm);
        const double dx = (sm[0] - sm[2]) * a.inv_n, dy = (sm[1] - sm[3]) * a.inv_n;
        nh_spread = a.spread_nh * fast_sqrt(dx * dx + dy * dy);
      }
    }
  }

  // ---- nullhypo (IIF addFactor!(…, nullhypo=p), test/testPose3Pose3NH.jl:118): with probability p the factor does
  //      not apply to a particle; such particles keep their value and receive spreadNH · std entropy instead
  bool nullh[PPL];
#pragma unroll
  for (int k = 0; k < PPL; ++k) nullh[k] = false;
  double nh0_spread = 0.0;
  const double p_null = (!LEAN && a.nullhypo) ? a.nullhypo[c] : 0.0;
  if (p_null > 0.0) {  // wave-uniform
    const double sd0 = FP::template spread<PPL>(t, aux, act, a.inv_n, a.inv_nm1);
    nh0_spread = N > 1 ? a.spread_nh * (sd0 > 1e-10 ? sd0 : 1.0) : 0.0;   // calcStdBasicSpread fallback, as the inflation spread
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      const uint32_t ii = (uint32_t)(act[k] ? slot_particle<PPL>(lane, k) : 0);
      const u32x4 w0 = philox4x32_10(u32x4{ii, (uint32_t)stream, (uint32_t)(stream >> 32), (5u << 16)}, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
      nullh[k] = ((double)w0.x + 0.5) * (1.0 / 4294967296.0) < p_null;   // (the entropy words are re-drawn after the cycles:
    }                                                                      //  nothing but this flag stays live across the solve)
  }

  // Inflation cycles (IIF inflateCycles x {addEntropyOnManifold!, N x solve}) apply where the start point can reach the answer: Nelder-Mead
  // everywhere, every solver on the bearing-range pose direction (a ring of roots); the jitter is drawn exactly as the oracle defines
  // it (ro_rng_entropy).  On a unique-root factor CLOSED_FORM / NEWTON return the root and GAUSS_NEWTON iterates to it from the belief
  // point: one pass, no statistic, no entropy.
  const bool cyc_on = FP::needs_cycles(SOLVER, K);
  const int ncyc = cyc_on ? (a.cycles < 1 ? 1 : a.cycles) : 1;
  for (int cyc = 0; cyc < ncyc; ++cyc) {
    double spread = 0.0;
    if (cyc_on && a.inflation > 0.0 && N > 1) {
#ifdef ROME_EXPERIMENT_NO_SPREAD   // experiment build (scripts/br1_bounds.py): no cross-particle statistic at all -- the bound on what
      const double sd = a.inv_n * (double)N * 0.02;   // packing rows / cheaper reductions could ever save (a run-time constant)
#else
      const double sd = FP::template spread<PPL>(t, aux, act, a.inv_n, a.inv_nm1);
#endif
      spread = a.inflation * (sd > 1e-10 ? sd : 1.0);   // IIF calcStdBasicSpread: "if no std yet, set to 1"
    }
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      if (act[k] && sel[k] && !nullh[k]) {
        if (spread > 0.0) {
          double u[FP::DT];
#ifdef ROME_EXPERIMENT_NO_ENTROPY_RNG   // experiment build: the jitter without its Philox call (the bound on a cheaper generator)
#pragma unroll
          for (int d = 0; d < FP::DT; ++d) u[d] = 0.25 + 0.125 * (double)((lane + 3 * d + cyc) & 3);
#else
          rng_entropy_exact<FP::DT>(a.seed, stream, (uint32_t)slot_particle<PPL>(lane, k), cyc, u);
#endif
          FP::add_entropy(t[k], aux[k], spread, u);
        }
        st[k] = FP::template solve<SOLVER>(K, prep[k], z[k], fx[k], t[k], aux[k], a.max_iters, a.tol);
      }
    }
  }
  // NEWTON: the status is the residual FUNCTOR evaluated at the returned root (only when asked for)
  if constexpr (SOLVER == kSolverNewton) {
    if (a.status) {
#pragma unroll
      for (int k = 0; k < PPL; ++k)
        if (act[k] && sel[k] && !nullh[k]) st[k] = FP::verify(K, z[k], fx[k], t[k], aux[k], a.tol);
    }
  }

  if (p_null > 0.0 && nh0_spread > 0.0) {
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      if (act[k] && nullh[k]) {
        const uint32_t ii = (uint32_t)slot_particle<PPL>(lane, k);
        const u32x4 w0 = philox4x32_10(u32x4{ii, (uint32_t)stream, (uint32_t)(stream >> 32), (5u << 16)}, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        double u[FP::DT];
        const uint32_t e0[3] = {w0.y, w0.z, w0.w};
#pragma unroll
        for (int d = 0; d < (FP::DT < 3 ? FP::DT : 3); ++d) u[d] = ((double)e0[d] + 0.5) * (1.0 / 4294967296.0);
        if constexpr (FP::DT > 3) {
          const u32x4 w1 = philox4x32_10(u32x4{ii, (uint32_t)stream, (uint32_t)(stream >> 32), (5u << 16) | 1u}, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
          const uint32_t e1[3] = {w1.x, w1.y, w1.z};
#pragma unroll
          for (int d = 3; d < FP::DT; ++d) u[d] = ((double)e1[d - 3] + 0.5) * (1.0 / 4294967296.0);
        }
        FP::add_entropy(t[k], aux[k], nh0_spread, u);
      }
    }
  }
  if constexpr ((FP::kHypoDir == 0 || FP::kHypoDir == 2) && !LEAN) {
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
      if (act[k] && !sel[k]) {  // the other hypothesis holds for this particle: entropy only
        // the words of the hypothesis draw are re-drawn here (same Philox call) instead of staying live across the cycles: kept in
        // per-slot arrays they were parked in LDS by the compiler
        const u32x4 hw = philox4x32_10(u32x4{(uint32_t)slot_particle<PPL>(lane, k), (uint32_t)stream, (uint32_t)(stream >> 32), (4u << 16)},
                                       (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        t[k][0] += nh_spread * (((double)hw.y + 0.5) * (1.0 / 4294967296.0) - 0.5);
        t[k][1] += nh_spread * (((double)hw.z + 0.5) * (1.0 / 4294967296.0) - 0.5);
        if constexpr (FP::DT == 3) t[k][2] += nh_spread * (((double)hw.w + 0.5) * (1.0 / 4294967296.0) - 0.5);   // a pose: the heading too
      }
    }
  }
  const int mslot = (a.n_mirror > 0 || a.mirror_map) ? mirror_slot(a, c_raw) : -1;   // wave-uniform
  double* mb = (mslot >= 0 && valid) ? a.mirror_out + (size_t)mslot * FP::DT * N : nullptr;
#pragma unroll
  for (int k = 0; k < PPL; ++k) {
    const int i = slot_particle<PPL>(lane, k);
    if (act[k] && valid) {
      FP::finalize(t[k], aux[k]);
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) store_stream(ob + d * N + i, t[k][d]);   // (not read again by this launch: written through)
      if (a.status) a.status[(size_t)c * N + i] = st[k];
      if (mb) {
#pragma unroll
        for (int d = 0; d < FP::DT; ++d) store_stream(mb + d * N + i, t[k][d]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_conv_flat -- the plain sweep of a UNIQUE-ROOT factor (Pose2Pose2 + PriorPose2 rows, bearing-range -> landmark, Pose3Pose3) with
// CLOSED_FORM / NEWTON and in-kernel noise: nothing couples the particles of a convolution (no inflation statistic), so the
// particles of consecutive table rows are packed densely onto the threads of a block:
//   thread  = one PAIR of neighbouring particles (2j, 2j+1) of one row: the two share a Box-Muller pair (D = 3) and are adjacent
//             in every SoA row -> one 16-byte load / store per coordinate;
//   block   = CPB consecutive rows x H = ceil(N/2) pairs  (N = 100: 5 rows x 50 = 250 of 256 threads, against 100 of 128
//             lane-slots with one wavefront per row);
//   per-factor constants (μ, chol Σ) are staged ONCE per block through LDS: the first NK threads of every row load one entry each of
//             their own row's factor, every thread reads its row's slot back (broadcast reads); the table row itself is a 16-byte
//             load per thread.
// The root comes from FP::prepare (the same function the wave-per-row kernel and the per-factor entry points use: bit-identical
// proposals); NEWTON additionally evaluates the residual functor at the root when a status array is asked for.
// Any N >= 2; the start points u0 are never read (48 B of HBM traffic per Pose2 particle: fixed 24 + proposal 24).
// ------------------------------------------------------------------------------------------
constexpr int kFlatThreads = 256;
constexpr int kFlatMaxRows = 16;    // rows per block (rows of >= 8 pair-threads: N >= 16)
template <class FP> struct FlatStage { static constexpr int kLanes = FP::NK <= 16 ? 16 : 32; };   // LDS doubles per row (>= NK)

#ifndef ROME_FLAT_MINWAVES
#define ROME_FLAT_MINWAVES 8   // Pose2 / Point2 sweeps: 8 waves per SIMD (<= 64 VGPRs)
#endif
template <class FP, int SOLVER, bool VERIFY, bool VEC2, int PP>
__device__ __forceinline__ void conv_flat_body(const ConvArgs& a, int H, int CPB, uint32_t magic, int blk, double* __restrict__ s_K);
#ifndef ROME_FLAT_PP
#define ROME_FLAT_PP 1   // neighbouring particle pairs per thread of the packed sweep (Pose2 / Point2 factors).  Measured: 2 pairs per
                          // thread (fewer, fatter waves, one generation) need 96 VGPRs + spills and run 21.8 µs against 8.0 µs: one pair it is
#endif
#ifndef ROME_FLAT_GN_MINWAVES
#define ROME_FLAT_GN_MINWAVES 4   // the functor-iterating packed sweep (Pose2 / Point2): <= 128 VGPRs
#endif
#ifndef ROME_FLAT_GN6_MINWAVES
#define ROME_FLAT_GN6_MINWAVES 3   // SE(3) functor iteration on unit quaternions (round 6): asked to fit 168 VGPRs (three waves per SIMD, 7 spilled registers)
                                   // 85.4 us on the 10k helix against 88.4 us at 188 VGPRs / two waves; the sched barrier between a thread's two particles stays.
                                   // (round 5, 3x3 frames, profiles/r05_p3p3_gn_one_particle.txt: ONE particle per thread measured slower, 165.5 against 155.8 us.)
#endif
#ifndef ROME_FLAT_CF6_MINWAVES
#define ROME_FLAT_CF6_MINWAVES 1   // SE(3) closed form: 132 VGPRs (three waves per SIMD) 48.5 us on the 10k helix; pinned to 128 (four waves, 20 B of scratch) 49.7 us
#endif
template <class FP, int SOLVER, bool VERIFY, bool VEC2, int PP>
__global__ void __launch_bounds__(kFlatThreads, FP::DT <= 3 ? ((SOLVER == kSolverClosedForm && !VERIFY) ? (PP == 1 ? ROME_FLAT_MINWAVES : 5) : ROME_FLAT_GN_MINWAVES)
                                                            : (SOLVER == kSolverGaussNewton ? ROME_FLAT_GN6_MINWAVES : (VERIFY ? 1 : ROME_FLAT_CF6_MINWAVES)))
k_conv_flat(const ConvArgs a, int H, int CPB, uint32_t magic) {
  __shared__ double s_K[kFlatMaxRows * (FlatStage<FP>::kLanes + 2)];
  conv_flat_body<FP, SOLVER, VERIFY, VEC2, PP>(a, H, CPB, magic, xcd_contiguous_block(blockIdx.x, gridDim.x), s_K);
}
template <class FP, int SOLVER, bool VERIFY, bool VEC2, int PP>
__device__ __forceinline__ void conv_flat_body(const ConvArgs& a, int H, int CPB, uint32_t magic, int blk, double* __restrict__ s_K) {
  constexpr int SLP = FlatStage<FP>::kLanes + 2;   // (+2: rows of a wave's two convolutions start in different banks)
  constexpr int NP = 2 * PP;                        // particles per thread: PP neighbouring pairs (2·PP consecutive particles)
  const int tid = threadIdx.x;
#ifdef ROME_FLAT_TRACE   // experiment build (scripts/flat_trace.py): per-block timestamps instead of the status array
  const uint64_t trace_t0 = wall_clock64();
#endif
  const int c0 = blk * CPB;
  const int N = a.N;
  // ---- this thread's (row, particle group)
  const int lc_raw = (int)(((uint32_t)tid * magic) >> 16);   // tid / H
  const int j = tid - lc_raw * H;
  const bool live = lc_raw < CPB && c0 + lc_raw < a.n_conv;
  const int lc = lc_raw < CPB ? lc_raw : CPB - 1;
  const int c = min(c0 + lc, a.n_conv - 1);
  const int4 row = *reinterpret_cast<const int4*>(a.rows4 + 4 * (size_t)c);
  const int dr = (FP::kHypoDir < 0 || FP::kHypoDir == 2) ? row.y : a.dir_all;
  const int i0 = NP * j;                      // particles i0 .. i0 + NP - 1 (the tail of a row may be shorter)
  // ---- per-factor constants -> LDS: the first threads of every row load one entry each of THEIR OWN row's factor (the factor
  //      index arrives with the row they need anyway: the load is issued beside the belief loads, nothing waits for it here;
  //      branch-free: every thread loads SOME valid entry, only the first NK of a row publish theirs)
  constexpr int KP = (FP::NK + 7) / 8;   // passes (H >= 8 threads per row)
  double kst[KP];
#pragma unroll
  for (int e = 0; e < KP; ++e) {
    const int q = min(j + e * H, FP::NK - 1);
    const double* src = q < FP::DZ ? a.mu + (size_t)FP::DZ * row.x + q : a.L + (size_t)FP::NL * row.x + (q - FP::DZ);
    kst[e] = *src;
  }
  const double* __restrict__ fb = a.bel_fixed + (size_t)row.z * FP::DF * N;
  double fx[NP][FP::DF];
  [[maybe_unused]] double t0[NP][FP::DT];   // GAUSS_NEWTON: the start points u0 (the target's current belief): +24 B per Pose2 particle
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int ip = i0 + 2 * p;
    if (VEC2) {   // (N even: a pair is inside the row or entirely beyond it)
      const int ii = ip < N ? ip : 0;
#pragma unroll
      for (int d = 0; d < FP::DF; ++d) {
        const double2 v = *reinterpret_cast<const double2*>(fb + (size_t)d * N + ii);
        fx[2 * p][d] = v.x; fx[2 * p + 1][d] = v.y;
      }
    } else {
#pragma unroll
      for (int d = 0; d < FP::DF; ++d) {
        fx[2 * p][d] = fb[(size_t)d * N + (ip < N ? ip : 0)]; fx[2 * p + 1][d] = fb[(size_t)d * N + (ip + 1 < N ? ip + 1 : 0)];
      }
    }
    if constexpr (SOLVER == kSolverGaussNewton) {
      const double* __restrict__ tb = a.bel_target + (size_t)row.w * FP::DT * N;
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) {
        t0[2 * p][d] = tb[(size_t)d * N + (ip < N ? ip : 0)]; t0[2 * p + 1][d] = tb[(size_t)d * N + (ip + 1 < N ? ip + 1 : 0)];
      }
    }
  }
  // ---- measurement noise (depends on the row id only).  The two compiler fences keep the order {loads issued} -> {Philox /
  //      Box-Muller} -> {first use of a loaded value}, so that the generator runs under the load latency (left alone, the
  //      compiler sinks the generator below the LDS write and its s_waitcnt vmcnt(0))
  asm volatile("" ::: "memory");
  const uint64_t stream = a.stream_offset + (uint64_t)c;
  double xi[NP][FP::DZ];
#pragma unroll
  for (int p = 0; p < PP; ++p) rng_normals_pair<FP::DZ>(a.seed, stream, (uint32_t)(i0 + 2 * p), xi[2 * p], xi[2 * p + 1]);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
#pragma unroll
    for (int d = 0; d < FP::DZ; ++d) asm volatile("" : "+v"(xi[k][d]) :: "memory");
  }
#pragma unroll
  for (int e = 0; e < KP; ++e) {
    const int q = j + e * H;
    if (lc_raw < CPB && q < FP::NK) s_K[lc * SLP + q] = kst[e];
  }
  __syncthreads();
  const typename FP::Consts K = FP::from_lds(s_K + lc * SLP, dr);
  double t[NP][FP::DT];
  int st[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    double z[FP::DZ];
    st[k] = 0;
    FP::measurement(K, xi[k], z);
    const typename FP::Prep P = FP::prepare(K, z, fx[k]);
    if constexpr (SOLVER == kSolverGaussNewton) {   // the numerical root-find on the residual functor, from the belief point
      // SE(3): one particle's iteration holds two 3x3 frames, Exp(z_ω), the update and the residual (~110 VGPRs): the two particles of
      // a thread run one AFTER the other (no interleaving across this point), or the allocation doubles and one wave per SIMD is left
      if constexpr (FP::DT == 6) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) t[k][d] = t0[k][d];
      FP::canonical(t[k]);
      typename FP::Aux A = FP::init_aux(t[k]);
      st[k] = FP::template solve<kSolverGaussNewton>(K, P, z, fx[k], t[k], A, a.max_iters, a.tol);
      FP::finalize(t[k], A);
    } else {
      typename FP::Aux A;
      FP::template solve<kSolverClosedForm>(K, P, z, fx[k], t[k], A, 0, 0.0);
      if constexpr (VERIFY) st[k] = FP::verify(K, z, fx[k], t[k], A, a.tol);   // NEWTON with a status array: the functor at the root
      FP::finalize(t[k], A);
    }
  }
#ifdef ROME_FLAT_TRACE
  const uint64_t trace_t1 = wall_clock64();
#endif
  if (!live) return;
  double* __restrict__ ob = a.out + (size_t)c * FP::DT * N;
  const int mslot = (a.n_mirror > 0 || a.mirror_map) ? mirror_slot(a, c) : -1;
  double* mb = mslot >= 0 ? a.mirror_out + (size_t)mslot * FP::DT * N : nullptr;
#pragma unroll
  for (int p = 0; p < PP; ++p) {
    const int ip = i0 + 2 * p;
    if (ip >= N) continue;
    const bool act1 = ip + 1 < N;               // (odd N: the last pair is a single particle)
    if (VEC2) {
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) {
        const double2 v = {t[2 * p][d], t[2 * p + 1][d]};
        // streaming stores: the proposals are not read again by this launch; written through, they are not left dirty in the L2
        // for the end-of-kernel write-back (measured: 9.1 -> 7.8 µs per Manhattan sweep)
        store_stream2(ob + (size_t)d * N + ip, v);
        if (mb) store_stream2(mb + (size_t)d * N + ip, v);
      }
    } else {
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) {
        store_stream(ob + (size_t)d * N + ip, t[2 * p][d]); if (act1) store_stream(ob + (size_t)d * N + ip + 1, t[2 * p + 1][d]);
        if (mb) { store_stream(mb + (size_t)d * N + ip, t[2 * p][d]); if (act1) store_stream(mb + (size_t)d * N + ip + 1, t[2 * p + 1][d]); }
      }
    }
#ifndef ROME_FLAT_TRACE
    if (a.status) { a.status[(size_t)c * N + ip] = st[2 * p]; if (act1) a.status[(size_t)c * N + ip + 1] = st[2 * p + 1]; }
#endif
  }
#ifdef ROME_FLAT_TRACE
  if (a.status && tid == 0) {
    uint64_t* tr = reinterpret_cast<uint64_t*>(a.status) + 4 * (size_t)blockIdx.x;
    tr[0] = trace_t0; tr[1] = trace_t1; tr[2] = wall_clock64();
    tr[3] = (uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);   // HW_ID, XCC_ID
  }
#endif
}

// ------------------------------------------------------------------------------------------
// k_sweep_fused -- ONE launch for the whole sweep of a Pose2 / Point2 graph (MIT- / beehive-shaped: odometry + bearing-range
// sightings): the block range selects the family -- [bearing-range -> pose rows, one wavefront each | Pose2Pose2 + PriorPose2 rows,
// packed | bearing-range -> landmark rows, packed] -- and runs the SAME body as the family's own kernel (bit-identical proposals).
// A sub-generation table (a few thousand sightings) does not fill the chip and pays a launch each; fused, the long bearing-range ->
// pose waves are dispatched first and the packed blocks fill the machine around them.  CLOSED_FORM / NEWTON without status, plain
// rows (no pre-sampled noise / multihypo / nullhypo), 64 < N <= 128; anything else takes the per-family launches.
// ------------------------------------------------------------------------------------------
struct FusedArgs {
  ConvArgs br1, p2p2, br0;
  int nb_br1, nb_p2p2, nb_br0;       // blocks per part (each a multiple of 8: block b runs on XCD b % 8)
  int H, CPB2, CPB0;                 // packed parts: pair-threads per row, rows per block
  uint32_t magic;
};
// SOLVER: kSolverClosedForm (CLOSED_FORM / NEWTON) or kSolverGaussNewton -- the functor-iterating root-find of all three families in ONE
// launch (round 5: as three launches the bearing-range tables of an MIT-shaped graph are sub-generation, VALU-busy 0.30 / 0.39)
template <bool VEC2, int SOLVER>
__global__ void __launch_bounds__(256) k_sweep_fused(const FusedArgs f) {
  __shared__ double s_K[kFlatMaxRows * (FlatStage<P2P2>::kLanes + 2)];
  const int b = blockIdx.x;
  if (b < f.nb_br1) conv_wave_body<BR<1>, SOLVER, 2, true>(f.br1, xcd_contiguous_block(b, f.nb_br1));
  else if (b < f.nb_br1 + f.nb_p2p2) conv_flat_body<P2P2, SOLVER, false, VEC2, 1>(f.p2p2, f.H, f.CPB2, f.magic, xcd_contiguous_block(b - f.nb_br1, f.nb_p2p2), s_K);
  else conv_flat_body<BR<0>, SOLVER, false, VEC2, 1>(f.br0, f.H, f.CPB0, f.magic, xcd_contiguous_block(b - f.nb_br1 - f.nb_p2p2, f.nb_br0), s_K);
}

// The same with `multihypo` / `nullhypo` columns on the bearing-range tables (the beehive of BASELINE configs[3]: ambiguous re-sightings,
// test/testMultimodalRangeBearing.jl:53): both sighting directions run the feature-complete wave-per-row body (the fractional
// hypotheses need statistics over a row's particles), the odometry table stays packed.  One launch instead of three for a graph whose
// tables are all sub-generation; bit-identical to the per-family launches.
template <bool VEC2>
__global__ void __launch_bounds__(256) k_sweep_fused_mh(const FusedArgs f) {
  __shared__ double s_K[kFlatMaxRows * (FlatStage<P2P2>::kLanes + 2)];
  const int b = blockIdx.x;
  if (b < f.nb_br1) conv_wave_body<BR<1>, kSolverClosedForm, 2, false>(f.br1, xcd_contiguous_block(b, f.nb_br1));
  else if (b < f.nb_br1 + f.nb_p2p2) conv_flat_body<P2P2, kSolverClosedForm, false, VEC2, 1>(f.p2p2, f.H, f.CPB2, f.magic, xcd_contiguous_block(b - f.nb_br1, f.nb_p2p2), s_K);
  else conv_wave_body<BR<0>, kSolverClosedForm, 2, false>(f.br0, xcd_contiguous_block(b - f.nb_br1 - f.nb_p2p2, f.nb_br0));
}

// ------------------------------------------------------------------------------------------
// N > 512: the same convolution with the particles walked in chunks of 128 instead of living in registers for the whole
// kernel.  Cycle by cycle: (1) the spread of ALL N current points (the start points u0 in cycle 0, the previous cycle's
// solutions afterwards -- they are re-read from the proposal block itself), (2) chunk by chunk: load, re-draw the measurement
// samples (counter-based: the same every time), jitter with the oracle's uniforms, solve, store.  One wavefront per convolution; multihypo / nullhypo rows are not served here (the launcher refuses them).
// ------------------------------------------------------------------------------------------
template <class FP, int SOLVER>
__global__ void __launch_bounds__(64 * ROME_WPB) k_conv_big(const ConvArgs a) {
  constexpr int PPL = 2;
  const int lane = threadIdx.x & 63;
  const int c_raw = __builtin_amdgcn_readfirstlane(xcd_contiguous_block(blockIdx.x, gridDim.x) * ROME_WPB + (int)(threadIdx.x >> 6));
  const bool valid = c_raw < a.n_conv;
  const int c = valid ? c_raw : a.n_conv - 1;
  const int N = a.N;
  int f, dr, fv, tv;
  if (a.rows4) {
    const int4 row = *reinterpret_cast<const int4*>(a.rows4 + 4 * (size_t)c);
    f = row.x; fv = row.z; tv = row.w; dr = (FP::kHypoDir < 0 || FP::kHypoDir == 2) ? row.y : a.dir_all;
  } else {
    f = a.factor ? a.factor[c] : c; dr = a.dir ? a.dir[c] : a.dir_all;
    fv = a.fixed_var ? a.fixed_var[c] : c; tv = a.target_var ? a.target_var[c] : c;
  }
  const typename FP::Consts K = FP::load(a, f, dr);
  const double* __restrict__ fb = a.bel_fixed + (size_t)fv * FP::DF * N;
  const double* tb = a.bel_target + (size_t)tv * FP::DT * N;
  double* ob = a.out + (size_t)c * FP::DT * N;
  const uint64_t stream = a.stream_offset + (uint64_t)(a.row_stream ? a.row_stream[c] : c);
  const int meas_blk = a.meas_block ? a.meas_block[c] : -1;
  const bool cyc_on = FP::needs_cycles(SOLVER, K);
  const int ncyc = cyc_on ? (a.cycles < 1 ? 1 : a.cycles) : 1;
  if (!valid) return;   // (nothing below synchronises across waves; surplus waves of the last block have no row)
  for (int cyc = 0; cyc < ncyc; ++cyc) {
    const double* cur = cyc == 0 ? tb : ob;
    double spread = 0.0;
    if (cyc_on && a.inflation > 0.0 && N > 1) {
      double t0[FP::DT];
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) t0[d] = cur[d * N];
      FP::canonical(t0);
      const typename FP::Aux a0 = FP::init_aux(t0);
      const typename FP::Ref ref = FP::make_ref(t0, a0);
      double sm[2 * FP::DT];
#pragma unroll
      for (int j = 0; j < 2 * FP::DT; ++j) sm[j] = 0.0;
      for (int i = lane; i < N; i += 64) {
        double t[FP::DT], dd[FP::DT];
#pragma unroll
        for (int d = 0; d < FP::DT; ++d) t[d] = cur[d * N + i];
        FP::canonical(t);
        const typename FP::Aux ax = FP::init_aux(t);
        FP::tangent(ref, t, ax, dd);
#pragma unroll
        for (int d = 0; d < FP::DT; ++d) { sm[2 * d] += dd[d]; sm[2 * d + 1] += dd[d] * dd[d]; }
      }
      wave_sum_n<2 * FP::DT>(sm);
      double var = 0.0;
#pragma unroll
      for (int d = 0; d < FP::DT; ++d) var += fmax(0.0, (sm[2 * d + 1] - sm[2 * d] * sm[2 * d] * a.inv_n) * a.inv_nm1);
      const double sd = fast_sqrt(var);
      spread = a.inflation * (sd > 1e-10 ? sd : 1.0);
    }
    for (int base = 0; base < N; base += 64 * PPL) {
      double fx[PPL][FP::DF], t[PPL][FP::DT], z[PPL][FP::DZ];
      typename FP::Aux aux[PPL];
      bool act[PPL];
      [[maybe_unused]] double xi_odd[FP::DZ];
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const int i = base + 2 * lane + k;
        act[k] = i < N;
        const int ii = act[k] ? i : 0;
#pragma unroll
        for (int d = 0; d < FP::DF; ++d) fx[k][d] = fb[d * N + ii];
#pragma unroll
        for (int d = 0; d < FP::DT; ++d) t[k][d] = cur[d * N + ii];
        double xi[FP::DZ];
        if (meas_blk >= 0) {
          const double* nb = a.meas_base + (size_t)meas_blk * FP::DZ * N;
#pragma unroll
          for (int d = 0; d < FP::DZ; ++d) xi[d] = nb[d * N + ii];
        } else if (a.noise) {
          const double* nb = a.noise + (size_t)c * FP::DZ * N;
#pragma unroll
          for (int d = 0; d < FP::DZ; ++d) xi[d] = nb[d * N + ii];
        } else {   // the neighbours 2j, 2j+1 draw from the same Philox calls (rng_normals_pair)
          if ((k & 1) == 0) rng_normals_pair<FP::DZ>(a.seed, stream, (uint32_t)i, xi, xi_odd);
          else {
#pragma unroll
            for (int d = 0; d < FP::DZ; ++d) xi[d] = xi_odd[d];
          }
        }
        if ((a.noise && a.noise_is_meas) || meas_blk >= 0) {
#pragma unroll
          for (int d = 0; d < FP::DZ; ++d) z[k][d] = xi[d];
        } else FP::measurement(K, xi, z[k]);
        FP::canonical(t[k]);
        aux[k] = FP::init_aux(t[k]);
      }
#pragma unroll
      for (int k = 0; k < PPL; ++k) {
        const int i = base + 2 * lane + k;
        if (act[k]) {
          const typename FP::Prep prep = FP::prepare(K, z[k], fx[k]);
          if (spread > 0.0) {
            double u[FP::DT];
            rng_entropy_exact<FP::DT>(a.seed, stream, (uint32_t)i, cyc, u);
            FP::add_entropy(t[k], aux[k], spread, u);
          }
          int st = FP::template solve<SOLVER>(K, prep, z[k], fx[k], t[k], aux[k], a.max_iters, a.tol);
          if constexpr (SOLVER == kSolverNewton) { if (a.status) st = FP::verify(K, z[k], fx[k], t[k], aux[k], a.tol); }
          FP::finalize(t[k], aux[k]);
#pragma unroll
          for (int d = 0; d < FP::DT; ++d) ob[d * N + i] = t[k][d];
          if (a.status) a.status[(size_t)c * N + i] = st;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the next cycle's spread pass reads what other lanes of this wave just wrote
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

// ---- prior sampling: out = coords(exp_ϵ(hat(μ + Lξ))) ; one wave per prior
template <int D, int PPL>
__global__ void __launch_bounds__(256) k_sample_prior(const ConvArgs a) {
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  if (c >= a.n_conv) return;
  const int N = a.N;
  const int f = a.rows4 ? a.rows4[4 * (size_t)c] : (a.factor ? a.factor[c] : c);   // (clique tables: the factor is column 0 of the row)
  constexpr int NL = D * (D + 1) / 2;
  const double* mu = a.mu + (size_t)D * f;
  const double* L = a.L + (size_t)NL * f;
  double* ob = a.out + (size_t)c * D * N;
  const uint64_t stream = a.stream_offset + (uint64_t)(a.row_stream ? a.row_stream[c] : c);
#pragma unroll(PPL <= 8 ? PPL : 1)
  for (int k = 0; k < PPL; ++k) {
    const int i = lane + 64 * k;
    if (i < N) {
      double xi[D], zc[D];
      if (a.noise) {
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = a.noise[(size_t)c * D * N + d * N + i];
      } else rng_normals<D>(a.seed, stream, (uint32_t)i, xi);
      int p = 0;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        double s = mu[r];
#pragma unroll
        for (int j = 0; j <= r; ++j) s += L[p++] * xi[j];
        zc[r] = s;
      }
      if constexpr (D == 3) zc[2] = wrap_pi(zc[2]);
      else if constexpr (D == 6) { Se3 P; se3_from_coords(zc, P); se3_to_coords(P, zc); }   // (D == 2: a Point2, the sample itself)
#pragma unroll
      for (int d = 0; d < D; ++d) ob[d * N + i] = zc[d];
    }
  }
}

// ---- residual-only kernels (rows of AoS coordinates), used by the KAT entry points
__global__ void k_residual_pose2pose2(int n, const double* z, const double* p, const double* q, double* r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Se2 P = se2_from_coords(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  const Se2 Q = se2_from_coords(q[3 * i], q[3 * i + 1], q[3 * i + 2]);
  double sz, cz; fast_sincos(z[3 * i + 2], &sz, &cz);
  double rr[3];
  residual_pose2pose2(z[3 * i], z[3 * i + 1], cz, sz, P, Q, rr);
  r[3 * i] = rr[0]; r[3 * i + 1] = rr[1]; r[3 * i + 2] = rr[2];
}
__global__ void k_residual_priorpose2(int n, const double* m, const double* p, double* r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Se2 M = se2_from_coords(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
  const Se2 P = se2_from_coords(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  double rr[3];
  residual_priorpose2(M, P, rr);
  r[3 * i] = rr[0]; r[3 * i + 1] = rr[1]; r[3 * i + 2] = rr[2];
}
// p_is_point: 0 -> p rows are coords (x,y,θ); 1 -> native points [tx,ty,R11,R21,R12,R22]
__global__ void k_residual_bearingrange(int n, const double* z, const double* p, int p_is_point, const double* l, double* r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Se2 P;
  if (p_is_point) { P.x = p[6 * i]; P.y = p[6 * i + 1]; P.c = p[6 * i + 2]; P.s = p[6 * i + 3]; }
  else P = se2_from_coords(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
  double rr[2];
  residual_bearingrange(z[2 * i], z[2 * i + 1], P, l[2 * i], l[2 * i + 1], rr);
  r[2 * i] = rr[0]; r[2 * i + 1] = rr[1];
}
// p,q rows are native points (12 doubles: t, R col-major) when pts != 0, else coords (6)
__global__ void k_residual_pose3pose3(int n, const double* z, const double* p, const double* q, int pts, double* r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Se3 P, Q;
  if (pts) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { P.t[k] = p[12 * i + k]; Q.t[k] = q[12 * i + k]; }
#pragma unroll
    for (int k = 0; k < 9; ++k) { P.R[k] = p[12 * i + 3 + k]; Q.R[k] = q[12 * i + 3 + k]; }
  } else { se3_from_coords(p + 6 * i, P); se3_from_coords(q + 6 * i, Q); }
  double Z[9], rr[6];
  so3_exp(z + 6 * i + 3, Z);
  residual_pose3pose3(z + 6 * i, Z, P, Q, rr);
#pragma unroll
  for (int k = 0; k < 6; ++k) r[6 * i + k] = rr[k];
}
__global__ void k_residual_priorpose3(int n, const double* m, const double* p, double* r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Se3 M, P;
  se3_from_coords(m + 6 * i, M); se3_from_coords(p + 6 * i, P);
  double rr[6];
  residual_priorpose3(M, P, rr);
#pragma unroll
  for (int k = 0; k < 6; ++k) r[6 * i + k] = rr[k];
}

// ---- native point layouts <-> coordinates (rows): Pose2 [tx,ty,R11,R21,R12,R22] <-> (x,y,θ);
//      Pose3 [t(3), R col-major(9)] <-> (t, ω).  vee(log(ϵ,p)) / exp_ϵ(hat c) of src/variables/VariableTypes.jl:35,47.
__global__ void k_points_to_coords(int n, int dim, const double* __restrict__ pts, double* __restrict__ c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dim == 3) {
    c[3 * i] = pts[6 * i]; c[3 * i + 1] = pts[6 * i + 1]; c[3 * i + 2] = atan2(pts[6 * i + 3], pts[6 * i + 2]);
  } else {
    c[6 * i] = pts[12 * i]; c[6 * i + 1] = pts[12 * i + 1]; c[6 * i + 2] = pts[12 * i + 2];
    double R[9], w[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = pts[12 * i + 3 + k];
    so3_log(R, w);
    c[6 * i + 3] = w[0]; c[6 * i + 4] = w[1]; c[6 * i + 5] = w[2];
  }
}
__global__ void k_coords_to_points(int n, int dim, const double* __restrict__ c, double* __restrict__ pts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dim == 3) {
    double s, co; fast_sincos(c[3 * i + 2], &s, &co);
    pts[6 * i] = c[3 * i]; pts[6 * i + 1] = c[3 * i + 1];
    pts[6 * i + 2] = co; pts[6 * i + 3] = s; pts[6 * i + 4] = -s; pts[6 * i + 5] = co;
  } else {
    pts[12 * i] = c[6 * i]; pts[12 * i + 1] = c[6 * i + 1]; pts[12 * i + 2] = c[6 * i + 2];
    double R[9];
    so3_exp(c + 6 * i + 3, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) pts[12 * i + 3 + k] = R[k];
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_points_to_coords(int n, int dim, const double* pts, double* c, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_points_to_coords, dim3((n + 255) / 256), dim3(256), 0, s, n, dim, pts, c);
  return hipGetLastError();
}
hipError_t launch_coords_to_points(int n, int dim, const double* c, double* pts, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_coords_to_points, dim3((n + 255) / 256), dim3(256), 0, s, n, dim, c, pts);
  return hipGetLastError();
}
template <class FP, int SOLVER, bool LEAN>
static hipError_t launch_ppl_v(const ConvArgs& a, hipStream_t s) {
  const int nb = (a.n_conv + ROME_WPB - 1) / ROME_WPB;
  if (nb == 0) return hipSuccess;
  if (a.N <= 64)       hipLaunchKernelGGL((k_conv<FP, SOLVER, 1, LEAN>), dim3(nb), dim3(64 * ROME_WPB), 0, s, a);
  else if (a.N <= 128) hipLaunchKernelGGL((k_conv<FP, SOLVER, 2, LEAN>), dim3(nb), dim3(64 * ROME_WPB), 0, s, a);
  else if (a.N <= 256) hipLaunchKernelGGL((k_conv<FP, SOLVER, 4, LEAN>), dim3(nb), dim3(64 * ROME_WPB), 0, s, a);
  else if (a.N <= 512) hipLaunchKernelGGL((k_conv<FP, SOLVER, 8, LEAN>), dim3(nb), dim3(64 * ROME_WPB), 0, s, a);
  else {   // particles walked in chunks (k_conv_big); rows with multihypo / nullhypo / mirrors stay on the register-resident kernels
    if (a.alt_var || a.nullhypo || a.n_mirror > 0 || a.mirror_map) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_conv_big<FP, SOLVER>), dim3(nb), dim3(64 * ROME_WPB), 0, s, a);
  }
  return hipGetLastError();
}
// the packed sweep (k_conv_flat): H = ceil(N/2) pair-threads per row, CPB rows per 256-thread block
template <class FP, int SOLVER>
static hipError_t launch_flat(const ConvArgs& a, hipStream_t s) {
  // PP neighbouring pairs per thread: Pose2 / Point2 rows of >= 64 particles take ROME_FLAT_PP, everything else one pair
  constexpr int PPC = FP::DT <= 3 ? ROME_FLAT_PP : 1;
  const bool verify = SOLVER == kSolverNewton && a.status != nullptr;
  const int pp = (PPC > 1 && a.N >= 64 && !verify && SOLVER != kSolverGaussNewton) ? PPC : 1;
  const int H = (a.N + 2 * pp - 1) / (2 * pp);
  int CPB = kFlatThreads / H;
  if (CPB > kFlatMaxRows) CPB = kFlatMaxRows;
  const uint32_t magic = (65536u + (uint32_t)H - 1u) / (uint32_t)H;   // tid / H == (tid * magic) >> 16 for tid < 256 (checked below)
  for (int t = 0; t < kFlatThreads; ++t) if ((int)(((uint32_t)t * magic) >> 16) != t / H) return hipErrorInvalidValue;
  const int nb = (a.n_conv + CPB - 1) / CPB;
  if (nb == 0) return hipSuccess;
  // 16-byte accesses need an even N (row starts stay 16-byte aligned) and 16-byte aligned arrays
  const bool vec2 = (a.N % 2 == 0) && (((uintptr_t)a.bel_fixed | (uintptr_t)a.out | (uintptr_t)a.mirror_out) % 16 == 0);
  // (the functor evaluation is a separate instantiation: compiled into the plain sweep it would pin its register allocation)
  constexpr int CF = kSolverClosedForm, GN = kSolverGaussNewton;
  if constexpr (SOLVER == kSolverGaussNewton) {   // the functor-iterating packed sweep
    if (vec2) hipLaunchKernelGGL((k_conv_flat<FP, GN, false, true, 1>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
    else      hipLaunchKernelGGL((k_conv_flat<FP, GN, false, false, 1>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
  } else if (verify) {
    if (vec2) hipLaunchKernelGGL((k_conv_flat<FP, CF, true, true, 1>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
    else      hipLaunchKernelGGL((k_conv_flat<FP, CF, true, false, 1>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
  } else if (pp == 1) {
    if (vec2) hipLaunchKernelGGL((k_conv_flat<FP, CF, false, true, 1>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
    else      hipLaunchKernelGGL((k_conv_flat<FP, CF, false, false, 1>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
  } else {
    if (vec2) hipLaunchKernelGGL((k_conv_flat<FP, CF, false, true, PPC>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
    else      hipLaunchKernelGGL((k_conv_flat<FP, CF, false, false, PPC>), dim3(nb), dim3(kFlatThreads), 0, s, a, H, CPB, magic);
  }
  return hipGetLastError();
}
template <class FP, int SOLVER>
static hipError_t launch_ppl(const ConvArgs& a, hipStream_t s) {
  const bool lean = a.rows4 != nullptr && a.noise == nullptr && a.alt_var == nullptr && a.nullhypo == nullptr && a.row_stream == nullptr &&
                    a.meas_block == nullptr;
  if constexpr (FP::kUniqueRoot && (SOLVER == kSolverClosedForm || SOLVER == kSolverNewton || SOLVER == kSolverGaussNewton)) {
    // plain sweep of a unique-root factor: the packed kernel (rows of >= 8 pair-threads; tiny N stays one wavefront per row) -- the
    // analytic root, or the Gauss-Newton iteration on the residual functor from the belief point (nothing couples the particles of a
    // row either way: no inflation statistic)
    if (lean && a.N >= 16 && (a.N + 1) / 2 <= kFlatThreads) return launch_flat<FP, SOLVER>(a, s);
  }
  // NEWTON without a status array IS the closed form on every factor here (unique roots; the bearing-range pose direction steps
  // exactly onto the ring member its start selects): one instantiation (the functor evaluation of the status path would otherwise
  // pin the register allocation of the plain launch)
  if constexpr (SOLVER == kSolverNewton) { if (!a.status) return launch_ppl<FP, kSolverClosedForm>(a, s); }
  return lean ? launch_ppl_v<FP, SOLVER, true>(a, s) : launch_ppl_v<FP, SOLVER, false>(a, s);
}
template <class FP>
static hipError_t launch_solver(const ConvArgs& a, int solver, hipStream_t s) {
  switch (solver) {
    case kSolverClosedForm:  return launch_ppl<FP, kSolverClosedForm>(a, s);
    case kSolverNewton:      return launch_ppl<FP, kSolverNewton>(a, s);
    case kSolverNelderMead:  return launch_ppl<FP, kSolverNelderMead>(a, s);
    case kSolverGaussNewton: return launch_ppl<FP, kSolverGaussNewton>(a, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_conv_pose2pose2(const ConvArgs& a, int solver, hipStream_t s) { return launch_solver<P2P2>(a, solver, s); }
hipError_t launch_conv_pose3pose3(const ConvArgs& a, int solver, hipStream_t s) { return launch_solver<P3P3>(a, solver, s); }
hipError_t launch_conv_bearingrange(const ConvArgs& a, int solver, hipStream_t s) {
  return a.dir_all == 0 ? launch_solver<BR<0>>(a, solver, s) : launch_solver<BR<1>>(a, solver, s);
}
static bool plain_rows(const ConvArgs& a) { return a.rows4 && !a.noise && !a.alt_var && !a.nullhypo && !a.status && !a.row_stream && !a.meas_block; }
static bool hypo_rows(const ConvArgs& a) { return a.rows4 && !a.noise && !a.status && !a.row_stream && !a.meas_block && (a.alt_var || a.nullhypo); }
// the whole sweep of a Pose2 / Point2 graph: fused into one launch when every family takes its plain kernel, else family by family
hipError_t launch_sweep_pose2(const ConvArgs* p2p2, const ConvArgs* br1, const ConvArgs* br0, int solver, hipStream_t s) {
  const int N = p2p2 ? p2p2->N : (br1 ? br1->N : (br0 ? br0->N : 0));
  const bool shape_ok = p2p2 && br1 && br0 && p2p2->n_conv > 0 && br1->n_conv > 0 && br0->n_conv > 0 &&
                        (solver == kSolverClosedForm || solver == kSolverNewton || solver == kSolverGaussNewton) && N > 64 && N <= 128 && br1->N == N && br0->N == N &&
                        br1->dir_all == 1 && br0->dir_all == 0;
  // sighting tables with multihypo / nullhypo columns: the fused launch with the feature-complete wave bodies for both directions
  const bool fusable_mh = shape_ok && solver != kSolverGaussNewton && plain_rows(*p2p2) && (hypo_rows(*br1) || plain_rows(*br1)) && (hypo_rows(*br0) || plain_rows(*br0)) &&
                          (hypo_rows(*br1) || hypo_rows(*br0));
  const bool fusable = shape_ok &&
                       plain_rows(*p2p2) && plain_rows(*br1) && plain_rows(*br0) &&
                       // the fused kernel runs every part at the register allocation of the bearing-range pose body (88 VGPRs, 5 waves
                       // per SIMD instead of the packed sweep's 8): worth two saved launches unless the odometry table is both huge and
                       // dominant
                       (p2p2->n_conv <= 100000 || 10 * (br1->n_conv + br0->n_conv) >= p2p2->n_conv);
  if (!fusable && !fusable_mh) {
    hipError_t e = hipSuccess;
    if (br1 && br1->n_conv > 0 && (e = launch_conv_bearingrange(*br1, solver, s)) != hipSuccess) return e;
    if (p2p2 && p2p2->n_conv > 0 && (e = launch_conv_pose2pose2(*p2p2, solver, s)) != hipSuccess) return e;
    if (br0 && br0->n_conv > 0 && (e = launch_conv_bearingrange(*br0, solver, s)) != hipSuccess) return e;
    return e;
  }
  FusedArgs f;
  f.br1 = *br1; f.p2p2 = *p2p2; f.br0 = *br0;
  f.H = (N + 1) / 2;
  const int cpb = kFlatThreads / f.H;
  f.CPB2 = cpb < kFlatMaxRows ? cpb : kFlatMaxRows; f.CPB0 = f.CPB2;
  f.magic = (65536u + (uint32_t)f.H - 1u) / (uint32_t)f.H;
  for (int t = 0; t < kFlatThreads; ++t) if ((int)(((uint32_t)t * f.magic) >> 16) != t / f.H) return hipErrorInvalidValue;
  auto up8 = [](int n) { return (n + 7) & ~7; };
  f.nb_br1 = up8((br1->n_conv + ROME_WPB - 1) / ROME_WPB);
  f.nb_p2p2 = up8((p2p2->n_conv + f.CPB2 - 1) / f.CPB2);
  f.nb_br0 = fusable_mh ? up8((br0->n_conv + ROME_WPB - 1) / ROME_WPB) : up8((br0->n_conv + f.CPB0 - 1) / f.CPB0);
  const uintptr_t al = (uintptr_t)p2p2->bel_fixed | (uintptr_t)p2p2->out | (uintptr_t)p2p2->mirror_out | (uintptr_t)br0->bel_fixed |
                       (uintptr_t)br0->out | (uintptr_t)br0->mirror_out;
  const bool vec2 = (N % 2 == 0) && (al % 16 == 0);
  const int nb = f.nb_br1 + f.nb_p2p2 + f.nb_br0;
  if (fusable_mh) {
    if (vec2) hipLaunchKernelGGL((k_sweep_fused_mh<true>), dim3(nb), dim3(256), 0, s, f);
    else      hipLaunchKernelGGL((k_sweep_fused_mh<false>), dim3(nb), dim3(256), 0, s, f);
  } else if (solver == kSolverGaussNewton) {
    if (vec2) hipLaunchKernelGGL((k_sweep_fused<true, kSolverGaussNewton>), dim3(nb), dim3(256), 0, s, f);
    else      hipLaunchKernelGGL((k_sweep_fused<false, kSolverGaussNewton>), dim3(nb), dim3(256), 0, s, f);
  } else {
    if (vec2) hipLaunchKernelGGL((k_sweep_fused<true, kSolverClosedForm>), dim3(nb), dim3(256), 0, s, f);
    else      hipLaunchKernelGGL((k_sweep_fused<false, kSolverClosedForm>), dim3(nb), dim3(256), 0, s, f);
  }
  return hipGetLastError();
}
template <int D>
static hipError_t launch_prior(const ConvArgs& a, hipStream_t s) {
  const int nb = (a.n_conv + 3) / 4;
  if (nb == 0) return hipSuccess;
  if (a.N <= 64)       hipLaunchKernelGGL((k_sample_prior<D, 1>), dim3(nb), dim3(256), 0, s, a);
  else if (a.N <= 128) hipLaunchKernelGGL((k_sample_prior<D, 2>), dim3(nb), dim3(256), 0, s, a);
  else if (a.N <= 256) hipLaunchKernelGGL((k_sample_prior<D, 4>), dim3(nb), dim3(256), 0, s, a);
  else if (a.N <= 512) hipLaunchKernelGGL((k_sample_prior<D, 8>), dim3(nb), dim3(256), 0, s, a);
  else if (a.N <= 4096) hipLaunchKernelGGL((k_sample_prior<D, 64>), dim3(nb), dim3(256), 0, s, a);   // (a runtime-bounded loop over 64 slots)
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_sample_priorpose2(const ConvArgs& a, hipStream_t s) { return launch_prior<3>(a, s); }
hipError_t launch_sample_priorpose3(const ConvArgs& a, hipStream_t s) { return launch_prior<6>(a, s); }
hipError_t launch_sample_priorpoint2(const ConvArgs& a, hipStream_t s) { return launch_prior<2>(a, s); }

// ---- store <-> blocks of a device buffer (the receive side of a frontier exchange / a contiguous download buffer): one 256-thread
//      block per belief
__global__ void __launch_bounds__(256) k_scatter_blocks(int N, const int4* __restrict__ ent, double* buf, long long stride,
                                                        double* d2, double* dpt, double* d3, int to_store) {
  const int4 e = ent[blockIdx.x];   // (dim, var, block, type)
  double* sv = (e.w == 0 ? d2 : (e.w == 1 ? dpt : d3)) + (size_t)e.y * e.x * N;
  double* bb = buf + (size_t)e.z * (size_t)stride;
  if (to_store) { for (int q = threadIdx.x; q < e.x * N; q += 256) sv[q] = bb[q]; }
  else { for (int q = threadIdx.x; q < e.x * N; q += 256) bb[q] = sv[q]; }
}
hipError_t launch_scatter_blocks(int n, int N, const int32_t* ent, const double* buf, int64_t stride, double* st2, double* st_pt, double* st3,
                                 hipStream_t s, int to_store) {
  if (n > 0) hipLaunchKernelGGL(k_scatter_blocks, dim3(n), dim3(256), 0, s, N, reinterpret_cast<const int4*>(ent), const_cast<double*>(buf), (long long)stride,
                                st2, st_pt, st3, to_store);
  return hipGetLastError();
}

// ---- block operations inside a store (copy / anchor / relative / compose): one 256-thread block per entry (type | flags << 8, a, b, dst)
__global__ void __launch_bounds__(256) k_block_ops(int op, int N, const int4* __restrict__ ent, double* d2, double* dpt, double* d3, const double* __restrict__ prm) {
  int4 e = ent[blockIdx.x];
  const int flags = e.x >> 8;   // (compose: bit 0 = take A^-1, bit 1 = take B^-1)
  e.x &= 0xff;
  const int dim = e.x == 0 ? 3 : (e.x == 1 ? 2 : 6);
  double* base = e.x == 0 ? d2 : (e.x == 1 ? dpt : d3);
  const double* A = base + (size_t)e.y * dim * N;
  double* D = base + (size_t)e.w * dim * N;
  const int i = threadIdx.x;
  if (op == 0) { for (int q = i; q < dim * N; q += 256) D[q] = A[q]; return; }
  if (op == 4) {   // mix: particle i of D <- particle i of A unless i % k == k - 1 (k = flags): D keeps every k-th particle of its own
    const int k = flags < 1 ? 1 : flags;
    for (int q = i; q < N; q += 256)
      if (q % k != k - 1)
        for (int d = 0; d < dim; ++d) D[(size_t)d * N + q] = A[(size_t)d * N + q];
    return;
  }
  if (op == 3) {   // compose, particle by particle, Pose2 coordinates (x, y, theta): D_i = A'_i (+) B'_i with A' = A or A^-1, B' = B or B^-1
    const double* Bq = d2 + (size_t)e.z * 3 * N;
    for (int q = i; q < N; q += 256) {
      double ax = A[q], ay = A[N + q], at = A[2 * N + q], bx = Bq[q], by = Bq[N + q], bt = Bq[2 * N + q];
      double sn, cs;
      if (flags & 1) { sincos(at, &sn, &cs); const double x = -(cs * ax + sn * ay), y = -(-sn * ax + cs * ay); ax = x; ay = y; at = -at; }
      if (flags & 2) { sincos(bt, &sn, &cs); const double x = -(cs * bx + sn * by), y = -(-sn * bx + cs * by); bx = x; by = y; bt = -bt; }
      sincos(at, &sn, &cs);
      double s2, c2; sincos(at + bt, &s2, &c2);
      D[q] = ax + cs * bx - sn * by; D[N + q] = ay + sn * bx + cs * by; D[2 * N + q] = atan2(s2, c2);
    }
    const double gt = prm ? prm[2 * (size_t)blockIdx.x] : 1.0, gth = prm ? prm[2 * (size_t)blockIdx.x + 1] : 1.0;
    if (gt == 1.0 && gth == 1.0) return;
    // inflate the deviations of the composed samples about their mean (star-mesh transform: the edge's spread grows by what the pair
    // shares with the other legs of the eliminated star): translation by gt, heading by gth.  Sums in a fixed order (as the anchor).
    __shared__ double rs[4][256];
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int q = i; q < N; q += 256) { double sn, cs; sincos(D[2 * N + q], &sn, &cs); a0 += D[q]; a1 += D[N + q]; a2 += sn; a3 += cs; }
    rs[0][i] = a0; rs[1][i] = a1; rs[2][i] = a2; rs[3][i] = a3;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (i < w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rs[k][i] += rs[k][i + w];
      }
      __syncthreads();
    }
    const double inv = 1.0 / (double)N, mx = rs[0][0] * inv, my = rs[1][0] * inv, mt = atan2(rs[2][0], rs[3][0]);
    for (int q = i; q < N; q += 256) {
      double sn, cs; sincos(D[2 * N + q] - mt, &sn, &cs);
      const double dt = atan2(sn, cs);
      double s2, c2; sincos(mt + gth * dt, &s2, &c2);
      D[q] = mx + gt * (D[q] - mx); D[N + q] = my + gt * (D[N + q] - my); D[2 * N + q] = atan2(s2, c2);
    }
    return;
  }
  if (op == 2) {   // relative to ref = particle 0 of the POSE2 block e.y: Pose2 -> tangent coordinates of ref^-1 * s_i; Point2 -> (bearing, range)
    const double* Rf = d2 + (size_t)e.y * 3 * N;
    const double* S = base + (size_t)e.z * dim * N;
    const double rx = Rf[0], ry = Rf[N], rt = Rf[2 * N];
    double sn, cs; sincos(rt, &sn, &cs);
    for (int q = i; q < N; q += 256) {
      const double dx = S[q] - rx, dy = S[N + q] - ry;
      const double lx = cs * dx + sn * dy, ly = -sn * dx + cs * dy;
      if (e.x == 0) {
        double s2, c2; sincos(S[2 * N + q] - rt, &s2, &c2);
        D[q] = lx; D[N + q] = ly; D[2 * N + q] = atan2(s2, c2);
      } else { D[q] = atan2(ly, lx); D[N + q] = sqrt(lx * lx + ly * ly); }
    }
    return;
  }
  // anchor: the mean point, N times.  Sums in a fixed order (lane partials -> LDS tree) so that the result does not depend on scheduling.
  __shared__ double red[8][256];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = i; q < N; q += 256) {
    if (e.x == 0) { double sn, cs; sincos(A[2 * N + q], &sn, &cs); acc[0] += A[q]; acc[1] += A[N + q]; acc[2] += sn; acc[3] += cs; }
    else if (e.x == 1) { acc[0] += A[q]; acc[1] += A[N + q]; }
    else { acc[0] += A[q]; acc[1] += A[N + q]; acc[2] += A[2 * N + q]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][i] = acc[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (i < w) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][i] += red[k][i + w];
    }
    __syncthreads();
  }
  const double inv = 1.0 / (double)N;
  double m[6];
  if (e.x == 0) { m[0] = red[0][0] * inv; m[1] = red[1][0] * inv; m[2] = atan2(red[2][0], red[3][0]); }
  else if (e.x == 1) { m[0] = red[0][0] * inv; m[1] = red[1][0] * inv; }
  else { m[0] = red[0][0] * inv; m[1] = red[1][0] * inv; m[2] = red[2][0] * inv; m[3] = A[3 * N]; m[4] = A[4 * N]; m[5] = A[5 * N]; }
  __syncthreads();
  for (int q = i; q < dim * N; q += 256) D[q] = m[q / N];
}
hipError_t launch_block_ops(int op, int n, int N, const int32_t* ent, double* st2, double* st_pt, double* st3, hipStream_t s, const double* prm) {
  if (n > 0) hipLaunchKernelGGL(k_block_ops, dim3(n), dim3(256), 0, s, op, N, reinterpret_cast<const int4*>(ent), st2, st_pt, st3, prm);
  return hipGetLastError();
}

static inline dim3 rows_grid(int n) { return dim3((n + 255) / 256); }
hipError_t launch_residual_pose2pose2(int n, const double* z, const double* p, const double* q, double* r, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_residual_pose2pose2, rows_grid(n), dim3(256), 0, s, n, z, p, q, r);
  return hipGetLastError();
}
hipError_t launch_residual_priorpose2(int n, const double* m, const double* p, double* r, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_residual_priorpose2, rows_grid(n), dim3(256), 0, s, n, m, p, r);
  return hipGetLastError();
}
hipError_t launch_residual_bearingrange(int n, const double* z, const double* p, int p_is_point, const double* l, double* r, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_residual_bearingrange, rows_grid(n), dim3(256), 0, s, n, z, p, p_is_point, l, r);
  return hipGetLastError();
}
hipError_t launch_residual_pose3pose3(int n, const double* z, const double* p, const double* q, int pts, double* r, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_residual_pose3pose3, rows_grid(n), dim3(256), 0, s, n, z, p, q, pts, r);
  return hipGetLastError();
}
hipError_t launch_residual_priorpose3(int n, const double* m, const double* p, double* r, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k_residual_priorpose3, rows_grid(n), dim3(256), 0, s, n, m, p, r);
  return hipGetLastError();
}

}  // namespace rome
