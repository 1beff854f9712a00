/* rome_oracle.c -- TEST INFRASTRUCTURE ONLY (see rome_oracle.h).
 *
 * Plain-C FP64 restatement of RoME.jl's factor residuals and of the
 * per-particle root-find loop IncrementalInference.jl runs around them inside
 * approxConvBelief.  `path:line` = /root/reference/path:line.
 *
 * PARITY PIN STATUS
 *   residuals ............ pinned by test/testParametricSimulated.jl:33-46,105-144,
 *                          test/testBearingRange2D.jl:44-253,
 *                          test/threeDimLinearProductTest.jl:150-167,
 *                          test/testPartialPose3.jl:390-436  (tests/golden/residual_kats.json)
 *   optimiser/RNG/entropy  PARITY UNPINNED by the reference (no seeds, no iterate checks);
 *                          restated from IIF 0.35 / Optim 1.x / Manifolds 0.10.1 (not vendored).
 *   convolution statistics pinned statistically on the reference's own solved graph examples/fg-after-solve.tar.gz
 *                          (tests/golden/manhattan500_reference_solve.npz, tests/test_gpu_reference_solve.py) and on the
 *                          statistical assertions of test/testBasicPose2Conv.jl, test/TestPoseAndPoint2Constraints.jl.
 */
#include "rome_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RO_PI 3.141592653589793238462643383279502884
#define RO_MAXN 6

/* ======================================================================== */
/* Philox4x32-10 (Salmon et al. 2011, Random123) -- integer, bit-exact        */
/* ======================================================================== */
void ro_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { RO_DOMAIN_NOISE = 1, RO_DOMAIN_ENTROPY = 2 };

static inline double u53(uint32_t hi, uint32_t lo) { /* (0,1] */
  uint64_t x = ((uint64_t)hi << 32) | lo;
  return (double)((x >> 11) + 1) * (1.0 / 9007199254740992.0);
}

/* d standard normals for (seed, stream, particle): Box-Muller on Philox words, two pairs per Philox call.
 * Replaces `rand(MvNormal)` of ⚠IIF sampleTangent / RoME getSample (src/factors/BearingRange2D.jl:17-27); the
 * reference's stream is unseeded -> unpinned.  The transform is DEFINED in IEEE single-precision operations (every
 * multiply-add an explicit fmaf, the file is built with -ffp-contract=off) so that the HIP path evaluates the same
 * function bit for bit up to its final double-precision square root and products:
 *   radius   u1 = x·2^-32, x = float(wa) + 1 in [1, 2^32]  (|n| <= 6.66 sigma);  -ln u1 = (32 - e) ln2 - ln m with
 *            x = m·2^e, m in [1,2), ln m by a degree-7 polynomial in m - 1.5 (|error| <= 2.7e-7);
 *   angle    a = (π/4)·int32(wb << 2)·2^-31 in [-π/4, π/4) from the low 30 bits of wb, (c, s) = (cos a, sin a) by the
 *            Cephes single-precision kernels; the direction (c - s, c + s)/√2 is the angle π/4 + a, uniform on the first
 *            quadrant, and bits 31 / 30 of wb mirror it into the other three;
 *   normals  n0 = ±√(-ln u1)·(c - s),  n1 = ±√(-ln u1)·(c + s)   (n0² + n1² = -2 ln u1 exactly as in Box-Muller).
 * The draws carry 24-bit mantissas in a double; their law differs from N(0,1) by < 1e-6 in Kolmogorov distance
 * (tests/test_host_logic.py::test_normal_generator_law).  FP64 Box-Muller cost the HIP path ~135 VALU instructions per
 * pair, this form ~50 (profiles/r02_*). */
static inline float ro_bits_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t ro_f32_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
static void ro_noise_words(uint64_t seed, uint64_t stream, uint32_t particle, uint32_t b, uint32_t w[4]) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {particle, (uint32_t)stream, (uint32_t)(stream >> 32), ((uint32_t)RO_DOMAIN_NOISE << 16) | b};
  ro_philox4x32_10(ctr, key, w);
}
void ro_box_muller(uint32_t wa, uint32_t wb, double* n0, double* n1) {
  const float x = (float)wa + 1.0f;
  const uint32_t xb = ro_f32_bits(x);
  const float ke = (float)(int32_t)(159u - (xb >> 23));                 /* 32 - e */
  const float t = ro_bits_f32((xb & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
  float p = 0x1.4fab76p-7f;
  p = fmaf(p, t, -0x1.1d4ffp-6f);
  p = fmaf(p, t, 0x1.a972ep-6f);
  p = fmaf(p, t, -0x1.90d3ap-5f);
  p = fmaf(p, t, 0x1.94a6a8p-4f);
  p = fmaf(p, t, -0x1.c72898p-3f);
  p = fmaf(p, t, 0x1.555544p-1f);
  p = fmaf(p, t, 0x1.9f324cp-2f);                                       /* ln m */
  const float h = fmaf(ke, 0x1.62e43p-1f, -p);                          /* -ln u1 (ln2 rounded to single) */
  const float rr = h > 0.0f ? sqrtf(h) : 0.0f;                             /* IEEE single-precision square root */
  const float a = (float)(int32_t)(wb << 2) * 0x1.921fb6p-32f;           /* (π/4)·2^-31 */
  const float z = a * a;
  float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf(z, sp, -1.6666654611e-1f);
  const float sn = fmaf(z * a, sp, a);
  float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf(z, cp, 4.166664568298827e-2f);
  const float cs = fmaf(z * z, cp, fmaf(z, -0.5f, 1.0f));
  const float m0 = rr * (cs - sn), m1 = rr * (cs + sn);                  /* single-precision products: the draws are floats */
  *n0 = (double)((wb & 0x80000000u) ? -m0 : m0);
  *n1 = (double)((wb & 0x40000000u) ? -m1 : m1);
}
/* Neighbouring particles 2j, 2j+1 draw from the SAME Philox calls, made with the EVEN particle id as counter -- the same rule as
 * rng_normals / rng_normals_pair in the HIP path, where one thread owns both particles (RNG stream unpinned by the reference).
 *   d = 2, 6: the words of calls b = 0 .. d/2-1 in order; the even particle takes words [0, d), the odd one [d, 2d); consecutive
 *             word pairs (radius word, angle word) -> one Box-Muller pair.
 *   d = 3:    ONE call for the six normals of the two particles, its 128 bits cut into six 21-bit fields:
 *             fields 0/1 = top 21 bits of words 0/1 -> normals 0, 1 of the even particle; fields 2/3 = top 21 bits of words 2/3 ->
 *             normals 0, 1 of the odd one; field 4 = low 11 bits of word 0 then bits 10..1 of word 1, field 5 = the same of
 *             words 2, 3 -> normal 2 of the even (first output) and the odd particle (second).  A field f enters ro_box_muller as
 *             the word (f << 11) | 0x400 (the centre of its bin).
 *   other d (not used by the supported factors): the particle's own calls, pairs in call order. */
void ro_rng_normals(uint64_t seed, uint64_t stream, uint32_t particle, int d, double* out) {
  const uint32_t pe = particle & ~1u;
  const int odd = (int)(particle & 1u);
  if (d == 3) {
    uint32_t w[4];
    double c, s;
    ro_noise_words(seed, stream, pe, 0, w);
    const uint32_t r1 = ((odd ? w[2] : w[0]) & 0xFFFFF800u) | 0x400u, a1 = ((odd ? w[3] : w[1]) & 0xFFFFF800u) | 0x400u;
    const uint32_t rc = (w[0] << 21) | ((w[1] & 0x7FEu) << 10) | 0x400u, ac = (w[2] << 21) | ((w[3] & 0x7FEu) << 10) | 0x400u;
    ro_box_muller(r1, a1, &out[0], &out[1]);
    ro_box_muller(rc, ac, &c, &s);
    out[2] = odd ? s : c;
    return;
  }
  if (d == 2 || d == 6) {
    uint32_t ww[12];
    for (int b = 0; b < d / 2; ++b) ro_noise_words(seed, stream, pe, (uint32_t)b, ww + 4 * b);
    for (int k = 0; k < d; k += 2) ro_box_muller(ww[odd * d + k], ww[odd * d + k + 1], &out[k], &out[k + 1]);
    return;
  }
  int ncall = (d + 3) / 4;
  for (int b = 0; b < ncall; ++b) {
    uint32_t w[4];
    ro_noise_words(seed, stream, particle, (uint32_t)b, w);
    for (int p = 0; p < 2 && 4 * b + 2 * p < d; ++p) {
      double n0, n1;
      ro_box_muller(w[2 * p], w[2 * p + 1], &n0, &n1);
      out[4 * b + 2 * p] = n0;
      if (4 * b + 2 * p + 1 < d) out[4 * b + 2 * p + 1] = n1;
    }
  }
}

/* d uniforms in (0,1) for the entropy inflation of cycle `cycle` (⚠IIF addEntropyOnManifold!:
 * spread·(rand(d) .- 0.5); RNG stream unpinned): one Philox call per particle and cycle (two for d = 6), one 32-bit word
 * per coordinate, u = (w + 0.5)/2^32.  Counter = (particle, stream, domain 2 | 2·cycle + block).
 * The HIP path draws exactly these in every kernel that jitters at all: Nelder-Mead (all factors) and every solver on the
 * bearing-range pose direction (a one-parameter family of roots: the start selects the member).  On the unique-root factors the
 * device's CLOSED_FORM / NEWTON return the analytic root and its GAUSS_NEWTON iterates from the UNJITTERED belief point without
 * inflation cycles -- no start point can move a converged unique root by more than the solver tolerance -- whereas this oracle's
 * Newton mode always runs every cycle with its jitter; the two are compared to <= 1e-9 (tests/test_gpu_parity.py). */
void ro_rng_entropy(uint64_t seed, uint64_t stream, uint32_t particle, int cycle, int d, double* out) {
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  for (int b = 0; 3 * b < d; ++b) {
    uint32_t ctr[4] = {particle, (uint32_t)stream, (uint32_t)(stream >> 32),
                       ((uint32_t)RO_DOMAIN_ENTROPY << 16) | (uint32_t)(2 * cycle + b)};
    uint32_t w[4];
    ro_philox4x32_10(ctr, key, w);
    for (int k = 3 * b; k < d && k < 3 * b + 3; ++k) out[k] = ((double)w[k - 3 * b] + 0.5) * (1.0 / 4294967296.0);
  }
}

/* ======================================================================== */
/* Manifold primitives                                    */
/* ======================================================================== */

/* ⚠Manifolds sym_rem(x) = (x ≈ π ? -π : rem(x, 2π, RoundNearest)); used at
 * src/factors/BearingRange2D.jl:60. isapprox default rtol = sqrt(eps). */
double ro_sym_rem(double x) {
  const double rtol = 1.4901161193847656e-8;
  double m = fabs(x) > RO_PI ? fabs(x) : RO_PI;
  if (fabs(x - RO_PI) <= rtol * m) return -RO_PI;
  return remainder(x, 2.0 * RO_PI);
}

/* getPoint(Pose2, c) = exp_ϵ(hat(c)) : ((x,y), R(θ)); layout src/variables/VariableTypes.jl:35 */
void ro_pose2_point_from_coords(const double c[3], double pt[6]) {
  double s = sin(c[2]), co = cos(c[2]);
  pt[0] = c[0]; pt[1] = c[1];
  pt[2] = co; pt[3] = s; pt[4] = -s; pt[5] = co; /* col-major [R11 R21 R12 R22] */
}
/* vee(log(ϵ, p)) : (x, y, atan2(R21, R11)) */
void ro_pose2_coords_from_point(const double pt[6], double c[3]) {
  c[0] = pt[0]; c[1] = pt[1]; c[2] = atan2(pt[3], pt[2]);
}

static inline void mat3_mul(const double* A, const double* B, double* C) { /* col-major */
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i)
      C[i + 3 * j] = A[i] * B[3 * j] + A[i + 3] * B[1 + 3 * j] + A[i + 6] * B[2 + 3 * j];
}
static inline void mat3_tmul(const double* A, const double* B, double* C) { /* C = Aᵀ B */
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i)
      C[i + 3 * j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[1 + 3 * j] + A[3 * i + 2] * B[2 + 3 * j];
}
static inline void mat3_vec(const double* A, const double* v, double* o) {
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[i + 3] * v[1] + A[i + 6] * v[2];
}

/* ⚠Manifolds exp!(::Rotations{3}, q, p=I, X): θ = ‖ω‖; a = sinθ/θ; b = (1-cosθ)/θ²; I + aX + bX². */
void ro_so3_exp(const double w[3], double R[9]) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double th = sqrt(th2);
  double a, b;
  if (th == 0.0) { a = 1.0; b = 0.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
  double x = w[0], y = w[1], z = w[2];
  /* X = [0 -z y; z 0 -x; -y x 0];  X² = w wᵀ - θ² I */
  R[0] = 1.0 + b * (x * x - th2); R[3] = -a * z + b * x * y;      R[6] = a * y + b * x * z;
  R[1] = a * z + b * x * y;       R[4] = 1.0 + b * (y * y - th2); R[7] = -a * x + b * y * z;
  R[2] = -a * y + b * x * z;      R[5] = a * x + b * y * z;       R[8] = 1.0 + b * (z * z - th2);
}

/* ⚠Manifolds log!(::Rotations{3}, X, p=I, U): cosθ=(tr U-1)/2; X = U/usinc_from_cos(cosθ) projected to skew;
 * cosθ ≈ -1 branch: axis of the +1 eigenvector times π (axis sign unpinned: chosen here from the
 * largest diagonal column of (U+I)/2 with the sign of the residual skew part when it is non-zero). */
void ro_so3_log(const double U[9], double w[3]) {
  double c = 0.5 * (U[0] + U[4] + U[8] - 1.0);
  double sx = U[5] - U[7], sy = U[6] - U[2], sz = U[1] - U[3]; /* (U - Uᵀ) vee: (U32-U23, U13-U31, U21-U12) */
  if (fabs(c + 1.0) <= 1.4901161193847656e-8) {
    double d0 = 0.5 * (U[0] + 1.0), d1 = 0.5 * (U[4] + 1.0), d2 = 0.5 * (U[8] + 1.0);
    double ax[3];
    if (d0 >= d1 && d0 >= d2) { ax[0] = d0; ax[1] = 0.25 * (U[1] + U[3]); ax[2] = 0.25 * (U[2] + U[6]); }
    else if (d1 >= d2)        { ax[0] = 0.25 * (U[1] + U[3]); ax[1] = d1; ax[2] = 0.25 * (U[5] + U[7]); }
    else                      { ax[0] = 0.25 * (U[2] + U[6]); ax[1] = 0.25 * (U[5] + U[7]); ax[2] = d2; }
    double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    double sgn = (ax[0] * sx + ax[1] * sy + ax[2] * sz) < 0.0 ? -1.0 : 1.0;
    double k = sgn * RO_PI / n;
    w[0] = k * ax[0]; w[1] = k * ax[1]; w[2] = k * ax[2];
    return;
  }
  double usinc; /* sinθ/θ from cosθ */
  if (c >= 1.0) usinc = 1.0;
  else if (c <= -1.0) usinc = 0.0;
  else usinc = sqrt(1.0 - c * c) / acos(c);
  double k = 0.5 / usinc;
  w[0] = k * sx; w[1] = k * sy; w[2] = k * sz;
}

void ro_pose3_point_from_coords(const double c[6], double pt[12]) {
  pt[0] = c[0]; pt[1] = c[1]; pt[2] = c[2];
  ro_so3_exp(c + 3, pt + 3);
}
void ro_pose3_coords_from_point(const double pt[12], double c[6]) {
  c[0] = pt[0]; c[1] = pt[1]; c[2] = pt[2];
  ro_so3_log(pt + 3, c + 3);
}

/* Rotations.jl RotXYZ(r,p,y) = Rx(r) Ry(p) Rz(y)  (test/testPartialPose3.jl:420) */
void ro_rotxyz(double r, double p, double y, double R[9]) {
  double cr = cos(r), sr = sin(r), cp = cos(p), sp = sin(p), cy = cos(y), sy = sin(y);
  double Rx[9] = {1, 0, 0, 0, cr, sr, 0, -sr, cr};
  double Ry[9] = {cp, 0, -sp, 0, 1, 0, sp, 0, cp};
  double Rz[9] = {cy, sy, 0, -sy, cy, 0, 0, 0, 1};
  double T[9];
  mat3_mul(Rx, Ry, T);
  mat3_mul(T, Rz, R);
}

/* ======================================================================== */
/* Residual functors on native points                                         */
/* ======================================================================== */

/* src/factors/Pose2D.jl:51-67 with _compose/_vee of src/factors/PriorPose2.jl:19-25.
 * X tangent container [Xt(2), skew(4)] -> θ = X.R[2,1];  p,q points [t(2), R col-major(4)]. */
void ro_residual_pose2pose2_pt(const double X[6], const double p[6], const double q[6], double r[3]) {
  /* ϵX = exp(M, ϵ0, X) :60 */
  double th = X[3];
  double s = sin(th), c = cos(th);
  double eXt0 = X[0], eXt1 = X[1];
  /* q̂ = _compose(M, p, ϵX) :62 -> (p.t + p.R ϵX.t , p.R ϵX.R) */
  double qh_t0 = p[0] + p[2] * eXt0 + p[4] * eXt1;
  double qh_t1 = p[1] + p[3] * eXt0 + p[5] * eXt1;
  double qh_R11 = p[2] * c + p[4] * s;
  double qh_R21 = p[3] * c + p[5] * s;
  /* X̂ = log(M, q, q̂) :63 -> (q̂.t - q.t , log_SO2(q.R, q̂.R)) ; U = q.Rᵀ q̂.R */
  double U11 = q[2] * qh_R11 + q[3] * qh_R21;
  double U21 = q[4] * qh_R11 + q[5] * qh_R21;
  /* _vee :65 */
  r[0] = qh_t0 - q[0];
  r[1] = qh_t1 - q[1];
  r[2] = atan2(U21, U11);
}

/* src/factors/PriorPose2.jl:37-47 : _vee(log(M, p, m)) */
void ro_residual_priorpose2_pt(const double m[6], const double p[6], double r[3]) {
  double U11 = p[2] * m[2] + p[3] * m[3];
  double U21 = p[4] * m[2] + p[5] * m[3];
  r[0] = m[0] - p[0];
  r[1] = m[1] - p[1];
  r[2] = atan2(U21, U11);
}

/* src/factors/BearingRange2D.jl:48-64. meas = [skew(4) col-major, ρ] -> b = measX.x[1][2] */
void ro_residual_pose2point2br_pt(const double meas[5], const double p[6], const double l[2], double r[2]) {
  double dx = l[0] - p[0], dy = l[1] - p[1];
  /* pl = transpose(p.R) * (l - p.t) :57 */
  double plx = p[2] * dx + p[3] * dy;
  double ply = p[4] * dx + p[5] * dy;
  r[0] = ro_sym_rem(meas[1] - atan2(ply, plx)); /* :60 */
  r[1] = meas[4] - sqrt(plx * plx + ply * ply); /* :61 */
}

/* src/factors/Pose3Pose3.jl:17-29 : q̂ = compose(p, exp_ϵ(X)); coords(log(q, q̂)) */
void ro_residual_pose3pose3_pt(const double X[12], const double p[12], const double q[12], double r[6]) {
  double w[3] = {X[3 + 5], X[3 + 6], X[3 + 1]}; /* vee of the skew part: (X32, X13, X21) */
  double E[9], Rh[9], U[9], t[3];
  ro_so3_exp(w, E);
  mat3_mul(p + 3, E, Rh);      /* q̂.R = p.R ϵX.R */
  mat3_vec(p + 3, X, t);       /* p.R ϵX.t        */
  mat3_tmul(q + 3, Rh, U);     /* q.Rᵀ q̂.R        */
  r[0] = p[0] + t[0] - q[0];
  r[1] = p[1] + t[1] - q[1];
  r[2] = p[2] + t[2] - q[2];
  ro_so3_log(U, r + 3);
}

/* src/factors/Pose3D.jl:15-19 : vee(M, p, log(M, p, m)) */
void ro_residual_priorpose3_pt(const double m[12], const double p[12], double r[6]) {
  double U[9];
  mat3_tmul(p + 3, m + 3, U);
  r[0] = m[0] - p[0]; r[1] = m[1] - p[1]; r[2] = m[2] - p[2];
  ro_so3_log(U, r + 3);
}

/* ---- batched on coordinates ---- */
static inline void se2_hat(const double z[3], double X[6]) { /* ⚠Manifolds hat: ((x,y),[0 -θ; θ 0]) */
  X[0] = z[0]; X[1] = z[1]; X[2] = 0.0; X[3] = z[2]; X[4] = -z[2]; X[5] = 0.0;
}
static inline void se3_hat(const double z[6], double X[12]) {
  X[0] = z[0]; X[1] = z[1]; X[2] = z[2];
  double x = z[3], y = z[4], w = z[5];
  X[3] = 0;  X[4] = w;  X[5] = -y;
  X[6] = -w; X[7] = 0;  X[8] = x;
  X[9] = y;  X[10] = -x; X[11] = 0;
}
void ro_residual_pose2pose2(int n, const double* z, const double* p, const double* q, double* r) {
  for (int i = 0; i < n; ++i) {
    double X[6], P[6], Q[6];
    se2_hat(z + 3 * i, X);
    ro_pose2_point_from_coords(p + 3 * i, P);
    ro_pose2_point_from_coords(q + 3 * i, Q);
    ro_residual_pose2pose2_pt(X, P, Q, r + 3 * i);
  }
}
void ro_residual_priorpose2(int n, const double* m, const double* p, double* r) {
  for (int i = 0; i < n; ++i) {
    double M[6], P[6];
    ro_pose2_point_from_coords(m + 3 * i, M);
    ro_pose2_point_from_coords(p + 3 * i, P);
    ro_residual_priorpose2_pt(M, P, r + 3 * i);
  }
}
void ro_residual_pose2point2br(int n, const double* z, const double* p, const double* l, double* r) {
  for (int i = 0; i < n; ++i) {
    double meas[5] = {0.0, z[2 * i], -z[2 * i], 0.0, z[2 * i + 1]};
    double P[6];
    ro_pose2_point_from_coords(p + 3 * i, P);
    ro_residual_pose2point2br_pt(meas, P, l + 2 * i, r + 2 * i);
  }
}
void ro_residual_pose3pose3(int n, const double* z, const double* p, const double* q, double* r) {
  for (int i = 0; i < n; ++i) {
    double X[12], P[12], Q[12];
    se3_hat(z + 6 * i, X);
    ro_pose3_point_from_coords(p + 6 * i, P);
    ro_pose3_point_from_coords(q + 6 * i, Q);
    ro_residual_pose3pose3_pt(X, P, Q, r + 6 * i);
  }
}
void ro_residual_priorpose3(int n, const double* m, const double* p, double* r) {
  for (int i = 0; i < n; ++i) {
    double M[12], P[12];
    ro_pose3_point_from_coords(m + 6 * i, M);
    ro_pose3_point_from_coords(p + 6 * i, P);
    ro_residual_priorpose3_pt(M, P, r + 6 * i);
  }
}

/* ======================================================================== */
/* helpers                                                                    */
/* ======================================================================== */

/* Σ = L Lᵀ, lower, row-packed (what Distributions' MvNormal unwhitens with). */
int ro_cholesky_lower(int d, const double* cov, double* Lp) {
  double L[RO_MAXN * RO_MAXN];
  memset(L, 0, sizeof(L));
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = cov[i * d + j];
      for (int k = 0; k < j; ++k) s -= L[i * d + k] * L[j * d + k];
      if (i == j) { if (s <= 0.0) return -1; L[i * d + i] = sqrt(s); }
      else L[i * d + j] = s / L[j * d + j];
    }
  int k = 0;
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) Lp[k++] = L[i * d + j];
  return 0;
}

/* ⚠IIF calcStdBasicSpread: per-coordinate std (n-1 normalisation) of the belief's tangent coordinates.
 * Unpinned by the reference; definition shared with the HIP path: the tangent coordinates are taken about
 * particle 0, d_i = vee(log(x_0, x_i)) (translations: differences; SO(2): wrap(θ_i-θ_0); SO(3):
 * Log(R_0ᵀR_i)), and std_k = sqrt(var_i(d_i[k])).  `mean` = x_0 ⊕ mean_i(d_i) (one Karcher step from x_0). */
static void moments(int N, const double* d, double* mean, double* sd) {
  double m = 0; for (int i = 0; i < N; ++i) m += d[i]; m /= N;
  double v = 0; for (int i = 0; i < N; ++i) { double e = d[i] - m; v += e * e; }
  *mean = m; *sd = sqrt(v / (N > 1 ? (double)(N - 1) : 1.0));
}
void ro_belief_spread_se2(int N, const double* x, const double* y, const double* th, double* mean3, double* std3) {
  double* d = (double*)calloc((size_t)(N > 0 ? N : 1), sizeof(double));
  double m;
  for (int i = 0; i < N; ++i) d[i] = x[i] - x[0];
  moments(N, d, &m, &std3[0]); mean3[0] = x[0] + m;
  for (int i = 0; i < N; ++i) d[i] = y[i] - y[0];
  moments(N, d, &m, &std3[1]); mean3[1] = y[0] + m;
  double s0 = sin(th[0]), c0 = cos(th[0]);
  for (int i = 0; i < N; ++i) { double s = sin(th[i]), c = cos(th[i]); d[i] = atan2(c0 * s - s0 * c, c0 * c + s0 * s); } /* log_SO2(R(th0), R(th_i)) */
  moments(N, d, &m, &std3[2]); mean3[2] = th[0] + m;
  free(d);
}
void ro_belief_spread_r2(int N, const double* x, const double* y, double* mean2, double* std2) {
  double* d = (double*)calloc((size_t)(N > 0 ? N : 1), sizeof(double));
  double m;
  for (int i = 0; i < N; ++i) d[i] = x[i] - x[0];
  moments(N, d, &m, &std2[0]); mean2[0] = x[0] + m;
  for (int i = 0; i < N; ++i) d[i] = y[i] - y[0];
  moments(N, d, &m, &std2[1]); mean2[1] = y[0] + m;
  free(d);
}
void ro_belief_spread_se3(int N, const double* blk, double* mean6, double* std6) {
  double* d = (double*)malloc(sizeof(double) * 6 * N);
  double R0[9];
  { double w0[3] = {blk[3 * N], blk[4 * N], blk[5 * N]}; ro_so3_exp(w0, R0); }
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 3; ++k) d[k * N + i] = blk[k * N + i] - blk[k * N];
    double w[3] = {blk[3 * N + i], blk[4 * N + i], blk[5 * N + i]}, R[9], U[9], l[3];
    ro_so3_exp(w, R); mat3_tmul(R0, R, U); ro_so3_log(U, l);
    for (int k = 0; k < 3; ++k) d[(3 + k) * N + i] = l[k];
  }
  double md[6];
  for (int k = 0; k < 6; ++k) moments(N, d + k * N, &md[k], &std6[k]);
  for (int k = 0; k < 3; ++k) mean6[k] = blk[k * N] + md[k];
  { double E[9], Rm[9]; ro_so3_exp(md + 3, E); mat3_mul(R0, E, Rm); ro_so3_log(Rm, mean6 + 3); }
  free(d);
}

/* ======================================================================== */
/* Nelder-Mead, Optim.jl defaults (⚠Optim: NelderMead(AdaptiveParameters(),   */
/* AffineSimplexer()); g_tol 1e-8)                         */
/* ======================================================================== */
static void nm_sortperm(int m, const double* f, int* order) { /* stable: ties by storage index */
  for (int i = 0; i < m; ++i) order[i] = i;
  for (int i = 1; i < m; ++i) {
    int k = order[i], j = i - 1;
    while (j >= 0 && f[order[j]] > f[k]) { order[j + 1] = order[j]; --j; }
    order[j + 1] = k;
  }
}
static void nm_centroid(int n, int m, double simplex[][RO_MAXN], int skip, double* c) {
  for (int k = 0; k < n; ++k) c[k] = 0.0;
  for (int i = 0; i < m; ++i) if (i != skip) for (int k = 0; k < n; ++k) c[k] += simplex[i][k];
  double inv = 1.0 / n;
  for (int k = 0; k < n; ++k) c[k] *= inv;
}
static double nm_objective(int n, int m, const double* f) { /* sqrt(var(f) * n/m), var corrected */
  double mean = 0; for (int i = 0; i < m; ++i) mean += f[i]; mean /= m;
  double v = 0; for (int i = 0; i < m; ++i) { double d = f[i] - mean; v += d * d; }
  v /= (m - 1);
  return sqrt(v * ((double)n / (double)m));
}

int ro_nelder_mead(int n, ro_cost_fn f, void* ctx, double* x, int max_iters, double g_tol, int* n_evals) {
  const int m = n + 1;
  const double alpha = 1.0, beta = 1.0 + 2.0 / n, gamma = 0.75 - 1.0 / (2.0 * n), delta = 1.0 - 1.0 / n;
  double simplex[RO_MAXN + 1][RO_MAXN], fs[RO_MAXN + 1];
  int order[RO_MAXN + 1];
  int evals = 0;
  /* AffineSimplexer(a=0.025, b=0.5) */
  for (int i = 0; i < m; ++i) for (int k = 0; k < n; ++k) simplex[i][k] = x[k];
  for (int j = 0; j < n; ++j) simplex[j + 1][j] = (1.0 + 0.5) * simplex[j + 1][j] + 0.025;
  for (int i = 0; i < m; ++i) { fs[i] = f(simplex[i], ctx); ++evals; }
  nm_sortperm(m, fs, order);
  double nm_x = nm_objective(n, m, fs);
  int iter = 0, converged = nm_x <= g_tol;
  double xc[RO_MAXN], xr[RO_MAXN], xt[RO_MAXN], xl[RO_MAXN];
  while (!converged && iter < max_iters) {
    ++iter;
    int shrink = 0;
    int ih = order[m - 1];
    nm_centroid(n, m, simplex, ih, xc);
    for (int k = 0; k < n; ++k) xl[k] = simplex[order[0]][k];
    double f_lowest = fs[order[0]], f_second = fs[order[n - 1]], f_highest = fs[ih];
    for (int k = 0; k < n; ++k) xr[k] = xc[k] + alpha * (xc[k] - simplex[ih][k]);
    double f_reflect = f(xr, ctx); ++evals;
    if (f_reflect < f_lowest) {
      for (int k = 0; k < n; ++k) xt[k] = xc[k] + beta * (xr[k] - xc[k]);
      double f_expand = f(xt, ctx); ++evals;
      if (f_expand < f_reflect) { memcpy(simplex[ih], xt, n * sizeof(double)); fs[ih] = f_expand; }
      else                      { memcpy(simplex[ih], xr, n * sizeof(double)); fs[ih] = f_reflect; }
      for (int i = m - 1; i >= 1; --i) order[i] = order[i - 1];
      order[0] = ih;
    } else if (f_reflect < f_second) {
      memcpy(simplex[ih], xr, n * sizeof(double)); fs[ih] = f_reflect;
      nm_sortperm(m, fs, order);
    } else {
      if (f_reflect < f_highest) { /* outside contraction */
        for (int k = 0; k < n; ++k) xt[k] = xc[k] + gamma * (xr[k] - xc[k]);
        double fo = f(xt, ctx); ++evals;
        if (fo < f_reflect) { memcpy(simplex[ih], xt, n * sizeof(double)); fs[ih] = fo; nm_sortperm(m, fs, order); }
        else shrink = 1;
      } else {                     /* inside contraction */
        for (int k = 0; k < n; ++k) xt[k] = xc[k] - gamma * (xr[k] - xc[k]);
        double fi = f(xt, ctx); ++evals;
        if (fi < f_highest) { memcpy(simplex[ih], xt, n * sizeof(double)); fs[ih] = fi; nm_sortperm(m, fs, order); }
        else shrink = 1;
      }
    }
    if (shrink) {
      for (int i = 1; i < m; ++i) {
        int o = order[i];
        for (int k = 0; k < n; ++k) simplex[o][k] = xl[k] + delta * (simplex[o][k] - xl[k]);
        fs[o] = f(simplex[o], ctx); ++evals;
      }
      nm_sortperm(m, fs, order);
    }
    nm_x = nm_objective(n, m, fs);
    converged = nm_x <= g_tol;
  }
  /* after_while!: best vertex, or the centroid of the n best if it is lower */
  nm_sortperm(m, fs, order);
  nm_centroid(n, m, simplex, order[m - 1], xc);
  double fc = f(xc, ctx); ++evals;
  int imin = 0; for (int i = 1; i < m; ++i) if (fs[i] < fs[imin]) imin = i;
  if (fc < fs[imin]) memcpy(x, xc, n * sizeof(double));
  else memcpy(x, simplex[imin], n * sizeof(double));
  if (n_evals) *n_evals = evals;
  return converged ? 0 : 1;
}

/* ======================================================================== */
/* per-particle solves (⚠IIF _solveLambdaNumeric for AbstractManifoldMinimize: */
/* minimise Σ r(exp_ϵ(hat Xc))² over identity-chart coords Xc, start vee(log(ϵ,u0))) */
/* ======================================================================== */
typedef struct { double z[3]; double fixed_pt[6]; int dir; } p2p2_ctx;
static double p2p2_cost(const double* xc, void* vctx) {
  const p2p2_ctx* c = (const p2p2_ctx*)vctx;
  double X[6], T[6], r[3];
  se2_hat(c->z, X);
  ro_pose2_point_from_coords(xc, T); /* p = exp_ϵ(hat Xc) */
  if (c->dir == 0) ro_residual_pose2pose2_pt(X, c->fixed_pt, T, r);
  else             ro_residual_pose2pose2_pt(X, T, c->fixed_pt, r);
  return r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
}
static void p2p2_resid_coords(const double z[3], const double fx[3], const double t[3], int dir, double r[3]) {
  if (dir == 0) ro_residual_pose2pose2(1, z, fx, t, r); else ro_residual_pose2pose2(1, z, t, fx, r);
}
/* SURVEY A.5 */
static void p2p2_closed(const double z[3], const double fx[3], int dir, double t[3]) {
  if (dir == 0) {
    double s = sin(fx[2]), c = cos(fx[2]);
    t[0] = fx[0] + c * z[0] - s * z[1]; t[1] = fx[1] + s * z[0] + c * z[1]; t[2] = fx[2] + z[2];
  } else {
    double th = fx[2] - z[2];
    double s = sin(th), c = cos(th);
    t[0] = fx[0] - (c * z[0] - s * z[1]); t[1] = fx[1] - (s * z[0] + c * z[1]); t[2] = th;
  }
}
static int p2p2_newton(const double z[3], const double fx[3], int dir, double t[3], int max_iters, double tol) {
  for (int it = 0; it < max_iters; ++it) {
    double r[3];
    p2p2_resid_coords(z, fx, t, dir, r);
    double m = fmax(fabs(r[0]), fmax(fabs(r[1]), fabs(r[2])));
    if (m <= tol) return 0;
    if (dir == 0) { t[0] += r[0]; t[1] += r[1]; t[2] += r[2]; }
    else {
      double s = sin(t[2]), c = cos(t[2]);
      double J13 = -s * z[0] - c * z[1], J23 = c * z[0] - s * z[1];
      double dth = -r[2];
      t[0] += -r[0] - J13 * dth; t[1] += -r[1] - J23 * dth; t[2] += dth;
    }
  }
  return 1;
}
static inline double wrap_pi(double th) { return atan2(sin(th), cos(th)); }

/* entropy: u0 ← u0 ∘ exp_ϵ(hat(spread·(U-½)))  (⚠IIF addEntropyOnManifold!, compose form) */
static void se2_add_entropy(double t[3], double spread, const double u[3]) {
  double ex = spread * (u[0] - 0.5), ey = spread * (u[1] - 0.5), et = spread * (u[2] - 0.5);
  double s = sin(t[2]), c = cos(t[2]);
  t[0] += c * ex - s * ey; t[1] += s * ex + c * ey; t[2] += et;
}

static inline int get_idx(const int32_t* a, int c) { return a ? a[c] : c; }

/* The scalar spread IIF scales by `inflation` / `spreadNH` (⚠IIF calcStdBasicSpread = Manifolds.std(M, pts)): the square root
 * of the corrected Fréchet variance Σ_i d(mean, x_i)² / (n-1).  On these product manifolds d² is the sum of the squared
 * tangent-coordinate differences, so it is the root of the summed per-coordinate variances returned by ro_belief_spread_*. */
static double frechet_std(const double* sd, int d) {
  double v = 0.0;
  for (int k = 0; k < d; ++k) v += sd[k] * sd[k];
  v = sqrt(v);
  /* ⚠IIF calcStdBasicSpread: "if no std yet, set to 1" (msst = 1e-10 < σ ? σ : 1.0): a belief whose particles coincide
   * (approxConv / initVariable on an all-zero start) still gets jittered, so the one-parameter root family of the
   * bearing-range pose direction is spread around the landmark instead of collapsing onto one ray. */
  return v > 1e-10 ? v : 1.0;
}

/* nullhypo draw for particle i of stream st: -> 1 if the factor does NOT apply; u[0..d-1] entropy uniforms
 * (Philox domain 5; block 0: word0 = selector, words 1-3 = u0..u2; block 1: words 0-2 = u3..u5) */
static int nullhypo_draw(const ro_opts* o, uint64_t st, uint32_t i, int d, double* u) {
  uint32_t key[2] = {(uint32_t)o->seed, (uint32_t)(o->seed >> 32)};
  uint32_t ctr[4] = {i, (uint32_t)st, (uint32_t)(st >> 32), (5u << 16)}, w[4];
  ro_philox4x32_10(ctr, key, w);
  int isnull = (((double)w[0] + 0.5) * (1.0 / 4294967296.0)) < o->nullhypo;
  for (int k = 0; k < d && k < 3; ++k) u[k] = ((double)w[1 + k] + 0.5) * (1.0 / 4294967296.0);
  if (d > 3) {
    ctr[3] = (5u << 16) | 1u;
    ro_philox4x32_10(ctr, key, w);
    for (int k = 3; k < d; ++k) u[k] = ((double)w[k - 3] + 0.5) * (1.0 / 4294967296.0);
  }
  return isnull;
}

/* With multihypo over the SECOND pose of the factor (IIF `addFactor!(fg, [:a; :b1; :b2], Pose2Pose2(...), multihypo=[1; w; 1-w])`;
 * the reference's own uses of multihypo are all on bearing-range factors, same rule as ro_conv_pose2point2br_mh): alt_var[c] >= 0
 * names the other candidate's belief, hypo_w[c] the probability of the row's own.  Row direction 1 (solve the first pose): per
 * particle the fixed pose is drawn from (own, alt).  Row direction 0 (solve this candidate): particles drawn for the other
 * candidate are not constrained by the factor -- they keep their value and receive entropy
 * spread_nh · ‖mean_xy(this) − mean_xy(alt)‖ · (U − ½) on every coordinate. */
int ro_conv_pose2pose2_mh(const ro_opts* o, int C, const int32_t* factor, const int32_t* dir,
                          const int32_t* fixed_var, const int32_t* target_var,
                          const double* mu, const double* L, const double* bel, const double* noise,
                          double* out, int32_t* status, const int32_t* alt_var, const double* hypo_w, double spread_nh) {
  const int N = o->n_particles;
  if (N <= 0 || C < 0) return -1;
  int cycles = o->inflate_cycles < 1 ? 1 : o->inflate_cycles;
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < C; ++c) {
    const int f = get_idx(factor, c), dr = dir ? dir[c] : 0;
    const double* fb = bel + (size_t)get_idx(fixed_var, c) * 3 * N;
    const double* tb = bel + (size_t)get_idx(target_var, c) * 3 * N;
    const double* m = mu + 3 * f; const double* Lf = L + 6 * f;
    double* ob = out + (size_t)c * 3 * N;
    double* zs = (double*)malloc(sizeof(double) * 3 * N);
    for (int i = 0; i < N; ++i) {
      double xi[3];
      if (noise) { xi[0] = noise[(size_t)c * 3 * N + i]; xi[1] = noise[(size_t)c * 3 * N + N + i]; xi[2] = noise[(size_t)c * 3 * N + 2 * N + i]; }
      else ro_rng_normals(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, 3, xi);
      zs[3 * i + 0] = m[0] + Lf[0] * xi[0];
      zs[3 * i + 1] = m[1] + Lf[1] * xi[0] + Lf[2] * xi[1];
      zs[3 * i + 2] = m[2] + Lf[3] * xi[0] + Lf[4] * xi[1] + Lf[5] * xi[2];
      ob[i] = tb[i]; ob[N + i] = tb[N + i]; ob[2 * N + i] = wrap_pi(tb[2 * N + i]); /* X0c = vee(log(ϵ,u0)) */
      if (status) status[(size_t)c * N + i] = 0;
    }
    const int av = (alt_var && hypo_w && dr != 2) ? alt_var[c] : -1;
    const int hd = dr == 1 ? 1 : 0;                       /* which side of the factor is fractional */
    const double* ab = av >= 0 ? bel + (size_t)av * 3 * N : NULL;
    unsigned char* sel = (unsigned char*)malloc(N);
    double* mhu = (double*)malloc(sizeof(double) * 3 * N);
    for (int i = 0; i < N; ++i) {
      sel[i] = 1;
      if (av >= 0) {
        uint32_t key[2] = {(uint32_t)o->seed, (uint32_t)(o->seed >> 32)};
        uint64_t stq = o->stream_offset + (uint64_t)c;
        uint32_t ctr[4] = {(uint32_t)i, (uint32_t)stq, (uint32_t)(stq >> 32), (4u << 16)}, w4[4];
        ro_philox4x32_10(ctr, key, w4);
        sel[i] = (((double)w4[0] + 0.5) * (1.0 / 4294967296.0)) < hypo_w[c];
        for (int k = 0; k < 3; ++k) mhu[3 * i + k] = ((double)w4[1 + k] + 0.5) * (1.0 / 4294967296.0);
      }
    }
    unsigned char* nullh = (unsigned char*)calloc(N, 1);
    double* nhu = (double*)malloc(sizeof(double) * 3 * N);
    double nh_spread = 0.0;
    if (o->nullhypo > 0.0) {
      double m3[3], s3[3];
      ro_belief_spread_se2(N, ob, ob + N, ob + 2 * N, m3, s3);
      nh_spread = N > 1 ? o->spread_nh * frechet_std(s3, 3) : 0.0;
      for (int i = 0; i < N; ++i) nullh[i] = (unsigned char)nullhypo_draw(o, o->stream_offset + (uint64_t)c, (uint32_t)i, 3, nhu + 3 * i);
    }
    if (o->solver == RO_SOLVER_CLOSED_FORM) {
      for (int i = 0; i < N; ++i) {
        if (nullh[i] || (hd == 0 && !sel[i])) continue;
        const double* fs = (hd == 1 && !sel[i]) ? ab : fb;
        double fx[3] = {fs[i], fs[N + i], fs[2 * N + i]}, t[3];
        p2p2_closed(zs + 3 * i, fx, dr, t);
        ob[i] = t[0]; ob[N + i] = t[1]; ob[2 * N + i] = wrap_pi(t[2]);
      }
    } else {
      for (int cyc = 0; cyc < cycles; ++cyc) {
        double spread = 0.0;
        if (o->inflation > 0.0 && N > 1) {
          double mean3[3], std3[3];
          ro_belief_spread_se2(N, ob, ob + N, ob + 2 * N, mean3, std3);
          spread = o->inflation * frechet_std(std3, 3);
        }
        for (int i = 0; i < N; ++i) {
          if (nullh[i] || (hd == 0 && !sel[i])) continue;
          const double* fs = (hd == 1 && !sel[i]) ? ab : fb;
          double fx[3] = {fs[i], fs[N + i], fs[2 * N + i]};
          double t[3] = {ob[i], ob[N + i], ob[2 * N + i]};
          if (spread > 0.0) {
            double u[3];
            ro_rng_entropy(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, cyc, 3, u);
            se2_add_entropy(t, spread, u);
            t[2] = wrap_pi(t[2]);
          }
          int st;
          if (o->solver == RO_SOLVER_NEWTON) st = p2p2_newton(zs + 3 * i, fx, dr, t, o->max_iters, o->tol);
          else {
            p2p2_ctx cx; memcpy(cx.z, zs + 3 * i, sizeof(cx.z)); cx.dir = dr;
            ro_pose2_point_from_coords(fx, cx.fixed_pt);
            st = ro_nelder_mead(3, p2p2_cost, &cx, t, o->max_iters, o->tol, NULL);
          }
          ob[i] = t[0]; ob[N + i] = t[1]; ob[2 * N + i] = wrap_pi(t[2]);
          if (status && st) status[(size_t)c * N + i] = st;
        }
      }
    }
    if (nh_spread > 0.0)
      for (int i = 0; i < N; ++i) if (nullh[i]) {
        double t[3] = {ob[i], ob[N + i], ob[2 * N + i]};
        se2_add_entropy(t, nh_spread, nhu + 3 * i);
        ob[i] = t[0]; ob[N + i] = t[1]; ob[2 * N + i] = wrap_pi(t[2]);
      }
    if (av >= 0 && hd == 0) {   /* means of the ORIGINAL target belief (start points) and of the other candidate's belief */
      double mx = 0, my = 0, ax = 0, ay = 0;
      for (int i = 0; i < N; ++i) { mx += tb[i]; my += tb[N + i]; ax += ab[i]; ay += ab[N + i]; }
      const double dxm = (mx - ax) / N, dym = (my - ay) / N;
      const double nh = spread_nh * sqrt(dxm * dxm + dym * dym);
      for (int i = 0; i < N; ++i) if (!sel[i]) {
        ob[i] += nh * (mhu[3 * i] - 0.5); ob[N + i] += nh * (mhu[3 * i + 1] - 0.5);
        ob[2 * N + i] = wrap_pi(ob[2 * N + i] + nh * (mhu[3 * i + 2] - 0.5));
      }
    }
    free(zs); free(nullh); free(nhu); free(sel); free(mhu);
  }
  return 0;
}
int ro_conv_pose2pose2(const ro_opts* o, int C, const int32_t* factor, const int32_t* dir,
                       const int32_t* fixed_var, const int32_t* target_var,
                       const double* mu, const double* L, const double* bel, const double* noise,
                       double* out, int32_t* status) {
  return ro_conv_pose2pose2_mh(o, C, factor, dir, fixed_var, target_var, mu, L, bel, noise, out, status, NULL, NULL, 0.0);
}

/* ---- PriorPose2: N samples exp_ϵ(hat(μ + Lξ)) (⚠IIF samplePoint; src/factors/PriorPose2.jl:13-17) ---- */
int ro_sample_priorpose2(const ro_opts* o, int C, const int32_t* factor, const double* mu, const double* L,
                         const double* noise, double* out) {
  const int N = o->n_particles;
#pragma omp parallel for
  for (int c = 0; c < C; ++c) {
    const int f = get_idx(factor, c);
    const double* m = mu + 3 * f; const double* Lf = L + 6 * f;
    double* ob = out + (size_t)c * 3 * N;
    for (int i = 0; i < N; ++i) {
      double xi[3];
      if (noise) { xi[0] = noise[(size_t)c * 3 * N + i]; xi[1] = noise[(size_t)c * 3 * N + N + i]; xi[2] = noise[(size_t)c * 3 * N + 2 * N + i]; }
      else ro_rng_normals(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, 3, xi);
      ob[i] = m[0] + Lf[0] * xi[0];
      ob[N + i] = m[1] + Lf[1] * xi[0] + Lf[2] * xi[1];
      ob[2 * N + i] = wrap_pi(m[2] + Lf[3] * xi[0] + Lf[4] * xi[1] + Lf[5] * xi[2]);
    }
  }
  return 0;
}

/* ---- PriorPoint2: N samples μ + Lξ of the landmark prior (src/factors/Point2D.jl:8-18; ⚠IIF samplePoint) ---- */
int ro_sample_priorpoint2(const ro_opts* o, int C, const int32_t* factor, const double* mu, const double* L,
                          const double* noise, double* out) {
  const int N = o->n_particles;
  for (int c = 0; c < C; ++c) {
    const int f = get_idx(factor, c);
    const double* m = mu + 2 * f; const double* Lf = L + 3 * f;
    double* ob = out + (size_t)c * 2 * N;
    for (int i = 0; i < N; ++i) {
      double xi[2];
      if (noise) { xi[0] = noise[(size_t)c * 2 * N + i]; xi[1] = noise[(size_t)c * 2 * N + N + i]; }
      else ro_rng_normals(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, 2, xi);
      ob[i] = m[0] + Lf[0] * xi[0];
      ob[N + i] = m[1] + Lf[1] * xi[0] + Lf[2] * xi[1];
    }
  }
  return 0;
}

/* ---- Pose2Point2BearingRange ---- */
typedef struct { double z[2]; double fixed[3]; int dir; } br_ctx;
static void br_resid(const double z[2], const double pose[3], const double l[2], double r[2]) {
  ro_residual_pose2point2br(1, z, pose, l, r);
}
static double br_cost(const double* xc, void* vctx) {
  const br_ctx* c = (const br_ctx*)vctx;
  double r[2];
  if (c->dir == 0) br_resid(c->z, c->fixed, xc, r);       /* target = landmark (TranslationGroup(2): exp_ϵ(hat Xc) = Xc) */
  else             br_resid(c->z, xc, c->fixed, r);       /* target = pose, fixed = landmark (fixed[0..1]) */
  return r[0] * r[0] + r[1] * r[1];
}
static int br_newton(const double z[2], const double* fx, int dir, double* t, int max_iters, double tol) {
  for (int it = 0; it < max_iters; ++it) {
    double r[2];
    if (dir == 0) br_resid(z, fx, t, r); else br_resid(z, t, fx, r);
    if (fmax(fabs(r[0]), fabs(r[1])) <= tol) return 0;
    if (dir == 0) {
      /* Newton step in the pose-frame polar chart of the landmark: (φ, n) += (r0, r1) */
      double s = sin(fx[2]), c = cos(fx[2]);
      double dx = t[0] - fx[0], dy = t[1] - fx[1];
      double plx = c * dx + s * dy, ply = -s * dx + c * dy;
      double n = sqrt(plx * plx + ply * ply), phi = atan2(ply, plx);
      double nn = n + r[1], a = phi + r[0];
      double qx = nn * cos(a), qy = nn * sin(a);
      t[0] = fx[0] + c * qx - s * qy; t[1] = fx[1] + s * qx + c * qy;
    } else {
      /* under-determined (2 eq / 3 unknowns): exact block step that keeps the ray landmark->pose --
       * move along the ray to the measured range, then rotate to the measured bearing (for ranges >> 1 this is
       * the minimum-norm Gauss-Newton step, which spends the bearing error on the heading) */
      double dx = fx[0] - t[0], dy = fx[1] - t[1];
      double n = sqrt(dx * dx + dy * dy);
      double ux = n > 0 ? dx / n : 1.0, uy = n > 0 ? dy / n : 0.0;
      t[0] = fx[0] - z[1] * ux; t[1] = fx[1] - z[1] * uy;
      t[2] = atan2(uy, ux) - z[0];
    }
  }
  return 1;
}
static void br_closed(const double z[2], const double* fx, int dir, double* t) {
  if (dir == 0) { /* l = p.t + ρ R(θ)(cos b, sin b) */
    double a = fx[2] + z[0];
    t[0] = fx[0] + z[1] * cos(a); t[1] = fx[1] + z[1] * sin(a);
  } else {        /* member of the ring nearest (in translation) to the start point t */
    double dx = fx[0] - t[0], dy = fx[1] - t[1];
    double n = sqrt(dx * dx + dy * dy);
    double ux = n > 0 ? dx / n : 1.0, uy = n > 0 ? dy / n : 0.0;
    t[0] = fx[0] - z[1] * ux; t[1] = fx[1] - z[1] * uy;
    t[2] = atan2(uy, ux) - z[0];
  }
}

/* multihypo (⚠IIF computeAcrossHypothesis!, `multihypo=[1, w, 1-w]` over two landmark candidates,
 * test/testMultimodalRangeBearing.jl:53): alt_var[c] = the other landmark (-1 none), hypo_w[c] = probability of
 * the row's own landmark.  Per particle a categorical draw (Philox domain 4, word 0) picks the hypothesis.
 * dir 1: the fixed landmark particle is taken from the drawn landmark.  dir 0: particles of the other hypothesis
 * are not solved; after the cycles they receive entropy spread_nh · ‖mean(target) - mean(alt)‖ · (U-½)
 * (words 1,2 of the same call).  Unpinned by the reference; definition shared with the HIP path. */
int ro_conv_pose2point2br_mh(const ro_opts* o, int C, const int32_t* factor, int dir,
                             const int32_t* fixed_var, const int32_t* target_var,
                             const double* mu, const double* sigma,
                             const double* bel_fixed, const double* bel_target,
                             const double* noise, double* out, int32_t* status,
                             const int32_t* alt_var, const double* hypo_w, double spread_nh);
int ro_conv_pose2point2br(const ro_opts* o, int C, const int32_t* factor, int dir,
                          const int32_t* fixed_var, const int32_t* target_var,
                          const double* mu, const double* sigma,
                          const double* bel_fixed, const double* bel_target,
                          const double* noise, double* out, int32_t* status) {
  return ro_conv_pose2point2br_mh(o, C, factor, dir, fixed_var, target_var, mu, sigma, bel_fixed, bel_target, noise, out, status,
                                  NULL, NULL, 3.0);
}
int ro_conv_pose2point2br_mh(const ro_opts* o, int C, const int32_t* factor, int dir,
                             const int32_t* fixed_var, const int32_t* target_var,
                             const double* mu, const double* sigma,
                             const double* bel_fixed, const double* bel_target,
                             const double* noise, double* out, int32_t* status,
                             const int32_t* alt_var, const double* hypo_w, double spread_nh) {
  const int N = o->n_particles;
  const int df = dir == 0 ? 3 : 2, dt = dir == 0 ? 2 : 3;
  int cycles = o->inflate_cycles < 1 ? 1 : o->inflate_cycles;
#pragma omp parallel for schedule(dynamic, 4)
  for (int c = 0; c < C; ++c) {
    const int f = get_idx(factor, c);
    const double* fb = bel_fixed + (size_t)get_idx(fixed_var, c) * df * N;
    const double* tb = bel_target + (size_t)get_idx(target_var, c) * dt * N;
    double* ob = out + (size_t)c * dt * N;
    double* zs = (double*)malloc(sizeof(double) * 2 * N);
    const int av = (alt_var && hypo_w) ? alt_var[c] : -1;
    unsigned char* sel = (unsigned char*)malloc(N);
    double* nhu = (double*)malloc(sizeof(double) * 2 * N);
    const double* ab = av >= 0 ? ((dir == 1 ? bel_fixed : bel_target) + (size_t)av * (dir == 1 ? df : dt) * N) : NULL;
    for (int i = 0; i < N; ++i) {
      sel[i] = 1;
      if (av >= 0) {
        uint32_t key[2] = {(uint32_t)o->seed, (uint32_t)(o->seed >> 32)};
        uint64_t stq = o->stream_offset + (uint64_t)c;
        uint32_t ctr[4] = {(uint32_t)i, (uint32_t)stq, (uint32_t)(stq >> 32), (4u << 16)}, w4[4];
        ro_philox4x32_10(ctr, key, w4);
        sel[i] = (((double)w4[0] + 0.5) * (1.0 / 4294967296.0)) < hypo_w[c];
        nhu[2 * i] = ((double)w4[1] + 0.5) * (1.0 / 4294967296.0); nhu[2 * i + 1] = ((double)w4[2] + 0.5) * (1.0 / 4294967296.0);
      }
    }
    for (int i = 0; i < N; ++i) {
      double xi[2];
      if (noise) { xi[0] = noise[(size_t)c * 2 * N + i]; xi[1] = noise[(size_t)c * 2 * N + N + i]; }
      else ro_rng_normals(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, 2, xi);
      /* rand(bearing), rand(range) src/factors/BearingRange2D.jl:23.  sigma >= 0: Normal(mu, sigma);
       * sigma < 0: Uniform(mu - |sigma|, mu + |sigma|) via the normal CDF of the same xi (u = erfc(-xi/√2)/2) */
      for (int k = 0; k < 2; ++k) {
        double sg = sigma[2 * f + k];
        zs[2 * i + k] = sg >= 0.0 ? mu[2 * f + k] + sg * xi[k] : mu[2 * f + k] - sg * (erfc(-xi[k] * 0.70710678118654752440) - 1.0);
      }
      for (int k = 0; k < dt; ++k) ob[k * N + i] = tb[k * N + i];
      if (dt == 3) ob[2 * N + i] = wrap_pi(ob[2 * N + i]);
      if (status) status[(size_t)c * N + i] = 0;
    }
    /* nullhypo (same rule as the Pose2Pose2 convolution): such particles skip the solve and receive
     * spread_nh · std(start belief) entropy afterwards (frechet_std) */
    unsigned char* nullh = (unsigned char*)calloc(N, 1);
    double* nhu0 = (double*)malloc(sizeof(double) * 3 * N);
    double nh0_spread = 0.0;
    if (o->nullhypo > 0.0) {
      if (dt == 3) { double m3[3], s3[3]; ro_belief_spread_se2(N, ob, ob + N, ob + 2 * N, m3, s3); nh0_spread = N > 1 ? o->spread_nh * frechet_std(s3, 3) : 0.0; }
      else         { double m2[2], s2[2]; ro_belief_spread_r2(N, ob, ob + N, m2, s2); nh0_spread = N > 1 ? o->spread_nh * frechet_std(s2, 2) : 0.0; }
      for (int i = 0; i < N; ++i) nullh[i] = (unsigned char)nullhypo_draw(o, o->stream_offset + (uint64_t)c, (uint32_t)i, dt, nhu0 + 3 * i);
    }
    int ncyc = (o->solver == RO_SOLVER_CLOSED_FORM && dir == 0) ? 1 : cycles;
    for (int cyc = 0; cyc < ncyc; ++cyc) {
      double spread = 0.0;
      if (o->inflation > 0.0 && N > 1 && !(o->solver == RO_SOLVER_CLOSED_FORM && dir == 0)) {
        if (dt == 3) { double m3[3], s3[3]; ro_belief_spread_se2(N, ob, ob + N, ob + 2 * N, m3, s3); spread = o->inflation * frechet_std(s3, 3); }
        else         { double m2[2], s2[2]; ro_belief_spread_r2(N, ob, ob + N, m2, s2); spread = o->inflation * frechet_std(s2, 2); }
      }
      for (int i = 0; i < N; ++i) {
        double fx[3] = {0, 0, 0}, t[3] = {0, 0, 0};
        if (dir == 0 && !sel[i]) continue;                       /* other hypothesis: not constrained by this factor */
        if (nullh[i]) continue;
        for (int k = 0; k < df; ++k) fx[k] = (dir == 1 && !sel[i]) ? ab[k * N + i] : fb[k * N + i];
        for (int k = 0; k < dt; ++k) t[k] = ob[k * N + i];
        if (spread > 0.0) {
          double u[3];
          ro_rng_entropy(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, cyc, dt, u);
          if (dt == 3) { se2_add_entropy(t, spread, u); t[2] = wrap_pi(t[2]); }
          else { t[0] += spread * (u[0] - 0.5); t[1] += spread * (u[1] - 0.5); }
        }
        int st = 0;
        if (o->solver == RO_SOLVER_CLOSED_FORM) br_closed(zs + 2 * i, fx, dir, t);
        else if (o->solver == RO_SOLVER_NEWTON) st = br_newton(zs + 2 * i, fx, dir, t, o->max_iters, o->tol);
        else {
          br_ctx cx; cx.z[0] = zs[2 * i]; cx.z[1] = zs[2 * i + 1]; cx.dir = dir;
          cx.fixed[0] = fx[0]; cx.fixed[1] = fx[1]; cx.fixed[2] = fx[2];
          st = ro_nelder_mead(dt, br_cost, &cx, t, o->max_iters, o->tol, NULL);
        }
        for (int k = 0; k < dt; ++k) ob[k * N + i] = t[k];
        if (dt == 3) ob[2 * N + i] = wrap_pi(t[2]);
        if (status && st) status[(size_t)c * N + i] = st;
      }
    }
    if (nh0_spread > 0.0)
      for (int i = 0; i < N; ++i) if (nullh[i]) {
        if (dt == 3) {
          double t[3] = {ob[i], ob[N + i], ob[2 * N + i]};
          se2_add_entropy(t, nh0_spread, nhu0 + 3 * i);
          ob[i] = t[0]; ob[N + i] = t[1]; ob[2 * N + i] = wrap_pi(t[2]);
        } else { ob[i] += nh0_spread * (nhu0[3 * i] - 0.5); ob[N + i] += nh0_spread * (nhu0[3 * i + 1] - 0.5); }
      }
    free(nullh); free(nhu0);
    if (av >= 0 && dir == 0) {
      /* means use the ORIGINAL target belief (start points) and the alternative landmark's belief */
      double mx = 0, my = 0, ax = 0, ay = 0;
      for (int i = 0; i < N; ++i) { mx += tb[i]; my += tb[N + i]; ax += ab[i]; ay += ab[N + i]; }
      double dxm = (mx - ax) / N, dym = (my - ay) / N;
      double nh = spread_nh * sqrt(dxm * dxm + dym * dym);
      for (int i = 0; i < N; ++i) if (!sel[i]) {
        ob[i] += nh * (nhu[2 * i] - 0.5); ob[N + i] += nh * (nhu[2 * i + 1] - 0.5);
      }
    }
    free(zs); free(sel); free(nhu);
  }
  return 0;
}

/* ---- Pose3Pose3 ---- */
typedef struct { double X[12]; double fixed_pt[12]; int dir; } p3p3_ctx;
static double p3p3_cost(const double* xc, void* vctx) {
  const p3p3_ctx* c = (const p3p3_ctx*)vctx;
  double T[12], r[6];
  ro_pose3_point_from_coords(xc, T);
  if (c->dir == 0) ro_residual_pose3pose3_pt(c->X, c->fixed_pt, T, r);
  else             ro_residual_pose3pose3_pt(c->X, T, c->fixed_pt, r);
  double s = 0; for (int k = 0; k < 6; ++k) s += r[k] * r[k];
  return s;
}
static void p3p3_closed_pt(const double z[6], const double F[12], int dir, double T[12]) {
  double Z[9];
  ro_so3_exp(z + 3, Z);
  if (dir == 0) { /* R_q = R_p Exp(z_ω), q.t = p.t + R_p z_t */
    double v[3]; mat3_mul(F + 3, Z, T + 3); mat3_vec(F + 3, z, v);
    for (int k = 0; k < 3; ++k) T[k] = F[k] + v[k];
  } else {        /* R_p = R_q Exp(z_ω)ᵀ, p.t = q.t - R_p z_t */
    double Zt[9]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Zt[i + 3 * j] = Z[j + 3 * i];
    double v[3]; mat3_mul(F + 3, Zt, T + 3); mat3_vec(T + 3, z, v);
    for (int k = 0; k < 3; ++k) T[k] = F[k] - v[k];
  }
}
/* Newton on the group: right-perturbation updates that zero the residual */
static int p3p3_newton_pt(const double z[6], const double F[12], int dir, double T[12], int max_iters, double tol) {
  double X[12], Z[9];
  se3_hat(z, X); ro_so3_exp(z + 3, Z);
  for (int it = 0; it < max_iters; ++it) {
    double r[6];
    if (dir == 0) ro_residual_pose3pose3_pt(X, F, T, r); else ro_residual_pose3pose3_pt(X, T, F, r);
    double m = 0; for (int k = 0; k < 6; ++k) m = fmax(m, fabs(r[k]));
    if (m <= tol) return 0;
    if (dir == 0) {
      double E[9], Rn[9];
      ro_so3_exp(r + 3, E); mat3_mul(T + 3, E, Rn); memcpy(T + 3, Rn, sizeof(Rn));
      T[0] += r[0]; T[1] += r[1]; T[2] += r[2];
    } else {
      double d[3], E[9], Rn[9], v[3];
      mat3_vec(Z, r + 3, d); d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];   /* δ = -Z r_ω */
      ro_so3_exp(d, E); mat3_mul(T + 3, E, Rn); memcpy(T + 3, Rn, sizeof(Rn));
      mat3_vec(T + 3, z, v);
      for (int k = 0; k < 3; ++k) T[k] = F[k] - v[k];                     /* t_p ← t_q - R_p z_t */
    }
  }
  return 1;
}
static void se3_add_entropy_pt(double T[12], double spread, const double u[6]) {
  double e[6]; for (int k = 0; k < 6; ++k) e[k] = spread * (u[k] - 0.5);
  double E[9], Rn[9], v[3];
  mat3_vec(T + 3, e, v);
  T[0] += v[0]; T[1] += v[1]; T[2] += v[2];
  ro_so3_exp(e + 3, E); mat3_mul(T + 3, E, Rn); memcpy(T + 3, Rn, sizeof(Rn));
}

int ro_conv_pose3pose3(const ro_opts* o, int C, const int32_t* factor, const int32_t* dir,
                       const int32_t* fixed_var, const int32_t* target_var,
                       const double* mu, const double* L, const double* bel, const double* noise,
                       double* out, int32_t* status) {
  const int N = o->n_particles;
  int cycles = o->inflate_cycles < 1 ? 1 : o->inflate_cycles;
#pragma omp parallel for schedule(dynamic, 2)
  for (int c = 0; c < C; ++c) {
    const int f = get_idx(factor, c), dr = dir ? dir[c] : 0;
    const double* fb = bel + (size_t)get_idx(fixed_var, c) * 6 * N;
    const double* tb = bel + (size_t)get_idx(target_var, c) * 6 * N;
    const double* m = mu + 6 * f; const double* Lf = L + 21 * f;
    double* ob = out + (size_t)c * 6 * N;
    double* zs = (double*)malloc(sizeof(double) * 6 * N);
    for (int i = 0; i < N; ++i) {
      double xi[6];
      if (noise) for (int k = 0; k < 6; ++k) xi[k] = noise[(size_t)c * 6 * N + k * N + i];
      else ro_rng_normals(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, 6, xi);
      int p = 0;
      for (int k = 0; k < 6; ++k) { double s = m[k]; for (int j = 0; j <= k; ++j) s += Lf[p++] * xi[j]; zs[6 * i + k] = s; }
      /* X0c = vee(log(ϵ,u0)): canonicalise the stored rotation vector through Exp/Log */
      double c0[6], P[12];
      for (int k = 0; k < 6; ++k) c0[k] = tb[k * N + i];
      ro_pose3_point_from_coords(c0, P); ro_pose3_coords_from_point(P, c0);
      for (int k = 0; k < 6; ++k) ob[k * N + i] = c0[k];
      if (status) status[(size_t)c * N + i] = 0;
    }
    unsigned char* nullh = (unsigned char*)calloc(N, 1);
    double* nhu = (double*)malloc(sizeof(double) * 6 * N);
    double nh_spread = 0.0;
    if (o->nullhypo > 0.0) {
      double m6[6], s6[6]; ro_belief_spread_se3(N, ob, m6, s6);
      nh_spread = N > 1 ? o->spread_nh * frechet_std(s6, 6) : 0.0;
      for (int i = 0; i < N; ++i) nullh[i] = (unsigned char)nullhypo_draw(o, o->stream_offset + (uint64_t)c, (uint32_t)i, 6, nhu + 6 * i);
    }
    int ncyc = o->solver == RO_SOLVER_CLOSED_FORM ? 1 : cycles;
    for (int cyc = 0; cyc < ncyc; ++cyc) {
      double spread = 0.0;
      if (o->inflation > 0.0 && N > 1 && o->solver != RO_SOLVER_CLOSED_FORM) {
        double m6[6], s6[6]; ro_belief_spread_se3(N, ob, m6, s6);
        spread = o->inflation * frechet_std(s6, 6);
      }
      for (int i = 0; i < N; ++i) {
        double fx[6], t[6], F[12], T[12];
        if (nullh[i]) continue;
        for (int k = 0; k < 6; ++k) { fx[k] = fb[k * N + i]; t[k] = ob[k * N + i]; }
        ro_pose3_point_from_coords(fx, F);
        ro_pose3_point_from_coords(t, T);
        if (spread > 0.0) {
          double u[6];
          ro_rng_entropy(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, cyc, 6, u);
          se3_add_entropy_pt(T, spread, u);
        }
        int st = 0;
        if (o->solver == RO_SOLVER_CLOSED_FORM) p3p3_closed_pt(zs + 6 * i, F, dr, T);
        else if (o->solver == RO_SOLVER_NEWTON) st = p3p3_newton_pt(zs + 6 * i, F, dr, T, o->max_iters, o->tol);
        else {
          p3p3_ctx cx; se3_hat(zs + 6 * i, cx.X); memcpy(cx.fixed_pt, F, sizeof(F)); cx.dir = dr;
          double x0[6]; ro_pose3_coords_from_point(T, x0);
          st = ro_nelder_mead(6, p3p3_cost, &cx, x0, o->max_iters, o->tol, NULL);
          ro_pose3_point_from_coords(x0, T);
        }
        ro_pose3_coords_from_point(T, t);
        for (int k = 0; k < 6; ++k) ob[k * N + i] = t[k];
        if (status && st) status[(size_t)c * N + i] = st;
      }
    }
    if (nh_spread > 0.0)
      for (int i = 0; i < N; ++i) if (nullh[i]) {
        double t[6], T[12];
        for (int k = 0; k < 6; ++k) t[k] = ob[k * N + i];
        ro_pose3_point_from_coords(t, T);
        se3_add_entropy_pt(T, nh_spread, nhu + 6 * i);
        ro_pose3_coords_from_point(T, t);
        for (int k = 0; k < 6; ++k) ob[k * N + i] = t[k];
      }
    free(zs); free(nullh); free(nhu);
  }
  return 0;
}

int ro_sample_priorpose3(const ro_opts* o, int C, const int32_t* factor, const double* mu, const double* L,
                         const double* noise, double* out) {
  const int N = o->n_particles;
#pragma omp parallel for
  for (int c = 0; c < C; ++c) {
    const int f = get_idx(factor, c);
    const double* m = mu + 6 * f; const double* Lf = L + 21 * f;
    double* ob = out + (size_t)c * 6 * N;
    for (int i = 0; i < N; ++i) {
      double xi[6], zc[6], P[12];
      if (noise) for (int k = 0; k < 6; ++k) xi[k] = noise[(size_t)c * 6 * N + k * N + i];
      else ro_rng_normals(o->seed, o->stream_offset + (uint64_t)c, (uint32_t)i, 6, xi);
      int p = 0;
      for (int k = 0; k < 6; ++k) { double s = m[k]; for (int j = 0; j <= k; ++j) s += Lf[p++] * xi[j]; zc[k] = s; }
      ro_pose3_point_from_coords(zc, P); ro_pose3_coords_from_point(P, zc);
      for (int k = 0; k < 6; ++k) ob[k * N + i] = zc[k];
    }
  }
  return 0;
}

int ro_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void ro_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ======================================================================== */
/* KDE bandwidth by leave-one-out likelihood cross-validation (⚠AMP manikde! / ⚠KDE.jl kde!(pts) "lcv";   */
/* SURVEY §8(a) row a11, §8(f) row 1)                                                                    */
/* ======================================================================== */
/* The reference wraps every convolution result in `manikde!`, which picks ONE bandwidth per coordinate by
 * maximising the leave-one-out log-likelihood of a 1-D Gaussian KDE (Euclidean coordinates: KernelDensityEstimate
 * `ksize(·,"lcv")`, a golden-section search with relative tolerance 1e-2; Circular coordinates: AMP's naive
 * cross-validation with Optim's GoldenSection, i.e. to ~1e-8).  Both packages are unvendored; the rule below is
 * PINNED ON REFERENCE OUTPUT: the 361 × 3 bandwidths the reference stored with its solved Manhattan-500 graph
 * (tests/golden/manhattan500_reference_solve.npz `bandwidth`, next to the particles they were selected for) are
 * reproduced by it to <= 0.7 % on x, y (the reference's own 1 % stopping rule) and <= 1e-4 on θ -- including the
 * bimodal beliefs whose likelihood has two local maxima (tests/test_oracle_golden.py).
 *   D_ij   = x_i - x_j                      (circular: wrapped to [-π, π])
 *   LL(h)  = Σ_i log max(Σ_{j≠i} exp(-½ D_ij²/h²), 1e-300) - N log((N-1) h √(2π))
 *   minm   = max(min_{i≠j} |D_ij|, 1e-6);  maxm = max_i y_i - min_i y_i,  y_i = D_i0
 *   h      = golden-section minimiser of -LL on the bracket (2 minm/(N-1), (minm+maxm)/2, 2 maxm), Numerical-Recipes
 *            form, stop when |x3-x0| <= tol (|x1|+|x2|) -- the bracket Ihler's KDE toolbox / KDE.jl use.       */
static double lcv_wrap(double d) { return d - 6.283185307179586476925287 * rint(d * 0.15915494309189533576888); }
static double lcv_negll(int N, const double* x, int circular, double h) {
  const double a = -0.5 / (h * h);
  double ll = 0.0;
  for (int i = 0; i < N; ++i) {
    double S = 0.0;
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      double d = x[i] - x[j];
      if (circular) d = lcv_wrap(d);
      S += exp(a * d * d);
    }
    ll += log(fmax(S, 1e-300));
  }
  return -(ll - N * log((N - 1) * h * 2.50662827463100050241576528));
}
double ro_kde_bandwidth_lcv(int N, const double* x, int circular, double tol, int* n_evals) {
  if (n_evals) *n_evals = 0;
  if (N < 2) return 0.0;
  double minm = INFINITY, ymin = 0.0, ymax = 0.0;
  for (int i = 0; i < N; ++i) {
    double y = x[i] - x[0];
    if (circular) y = lcv_wrap(y);
    if (y < ymin) ymin = y;
    if (y > ymax) ymax = y;
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      double d = x[i] - x[j];
      if (circular) d = lcv_wrap(d);
      if (fabs(d) < minm) minm = fabs(d);
    }
  }
  minm = fmax(minm, 1e-6);
  const double maxm = fmax(ymax - ymin, minm);
  const double ax = 2.0 * minm / (N - 1), bx = 0.5 * (minm + maxm), cx = 2.0 * maxm;
  const double Cg = 0.38196601125010515180, Rg = 0.61803398874989484820;
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + Cg * (cx - bx); }
  else { x2 = bx; x1 = bx - Cg * (bx - ax); }
  double f1 = lcv_negll(N, x, circular, x1), f2 = lcv_negll(N, x, circular, x2);
  int ne = 2;
  while (fabs(x3 - x0) > tol * (fabs(x1) + fabs(x2)) && ne < 200) {
    if (f2 < f1) { x0 = x1; x1 = x2; x2 = Rg * x1 + Cg * x3; f1 = f2; f2 = lcv_negll(N, x, circular, x2); }
    else         { x3 = x2; x2 = x1; x1 = Rg * x2 + Cg * x0; f2 = f1; f1 = lcv_negll(N, x, circular, x1); }
    ++ne;
  }
  if (n_evals) *n_evals = ne;
  return f1 < f2 ? x1 : x2;
}
/* bel [V][dim][N] (SoA blocks) -> bw [V][dim]; coordinate k is circular when bit k of circular_mask is set. */
int ro_kde_bandwidths(int dim, int V, int N, const double* bel, uint32_t circular_mask, double tol_euclid, double tol_circular, double* bw) {
  if (dim < 1 || dim > 6 || N < 2) return -1;
#pragma omp parallel for schedule(dynamic, 4)
  for (int t = 0; t < V * dim; ++t) {
    const int k = t % dim, circ = (circular_mask >> k) & 1;
    bw[t] = ro_kde_bandwidth_lcv(N, bel + (size_t)t * N, circ, circ ? tol_circular : tol_euclid, NULL);
  }
  return 0;
}

/* Max-density point estimate of a belief, coordinate by coordinate (⚠IIF getKDEMax, the `max` -- and, for headings, the
 * `suggested` -- entry of a variable's PPE).  PINNED ON REFERENCE OUTPUT: with the stored bandwidths, the
 * rule below reproduces the `ppe.max` of all 361 x 3 coordinates saved in the reference's solved Manhattan-500 graph
 * (tests/golden/manhattan500_reference_solve.npz `ppe[:,1]`), which all lie exactly on this grid:
 *   r = max_i x_i - min_i x_i;  X_g = (min - extend r) + g (1 + 2 extend) r / (G-1),  g = 0..G-1   (G = 200, extend = 0.1)
 *   y_g = Σ_j exp(-½ ((X_g - x_j)/h)²)     -- the Euclidean marginal KDE, also for headings (the reference does not wrap here)
 *   result = X_g at the FIRST g attaining max_g y_g.                                                                   */
int ro_kde_max(int dim, int V, int N, const double* bel /*[V][dim][N]*/, const double* bw /*[V][dim]*/, int G, double extend,
               double* out /*[V][dim]*/) {
  if (dim < 1 || N < 1 || G < 2) return -1;
#pragma omp parallel for schedule(dynamic, 8)
  for (int t = 0; t < V * dim; ++t) {
    const double* x = bel + (size_t)t * N;
    double lo = x[0], hi = x[0];
    for (int i = 1; i < N; ++i) { if (x[i] < lo) lo = x[i]; if (x[i] > hi) hi = x[i]; }
    const double r = hi - lo; lo -= extend * r; hi += extend * r;
    const double hb = fmax(bw[t], 1e-150), step = (hi - lo) / (G - 1), a = -0.5 / (hb * hb);
    double best = -1.0, xb = lo;
    for (int g = 0; g < G; ++g) {
      const double X = g == G - 1 ? hi : lo + g * step;
      double y = 0.0;
      for (int j = 0; j < N; ++j) { const double d = X - x[j]; y += exp(a * d * d); }
      if (y > best) { best = y; xb = X; }
    }
    out[t] = xb;
  }
  return 0;
}

/* ======================================================================== */
/* Product of proposals (stand-in for ⚠AMP manifoldProduct)          */
/* ======================================================================== */
/* NOT a restatement of ApproxManifoldProducts' multiscale Gibbs product (unvendored, unpinned): a
 * regularised importance-sampling product of the K proposal KDEs, defined once here and in the HIP path
 * (csrc/rome_product.hip):
 *   proposal l: N points, diagonal Gaussian kernels, bandwidth h_lk = max(c_N std_lk, 1e-6),
 *               c_N = (4/((d+2)N))^(1/(d+4)) (Silverman), std about particle 0 (ro_belief_spread_*)
 *   base b    : the proposal with the smallest Σ_k log h_lk (ties: lowest l) -- the tightest one
 *   candidates: the N points of b; log w_i = Σ_{l≠b} log Σ_j exp(-½ Σ_k ((x_i - y_lj)_k / h_lk)²)
 *   systematic resampling of N candidates with one uniform u (Philox domain 3), then kernel jitter
 *   x ⊕ (h_prod ⊙ ξ), 1/h_prod,k² = Σ_l 1/h_lk²  (ξ: ro_rng_normals on the variable's stream).
 *   K = 1: the proposal is copied; K = 0: the belief is kept.
 * dim = 2 (Point2), 3 (Pose2: third coordinate is an angle, differences wrapped) or 6 (Pose3: coordinates
 * [t; rotation vector]; tangent difference (x.t - y.t, Log(R_yᵀ R_x)), jitter R ← R Exp(h_ω ⊙ ξ_ω)).      */
static double wrap_diff(double a) { return atan2(sin(a), cos(a)); }

int ro_product_bw(const ro_opts* o, int dim, int V, const int32_t* prop_ptr, const int32_t* prop_rows,
                  const double* prop /*[rows][dim][N]*/, const double* prop_bw /*[rows][dim] or NULL: Silverman*/,
                  const double* bel_in /*[V][dim][N]*/, double* bel_out) {
  const int N = o->n_particles;
  if (dim != 2 && dim != 3 && dim != 6) return -1;
  const double cN = pow(4.0 / ((dim + 2.0) * N), 1.0 / (dim + 4.0));
#pragma omp parallel for schedule(dynamic, 8)
  for (int v = 0; v < V; ++v) {
    const int K = prop_ptr[v + 1] - prop_ptr[v];
    const int32_t* rows = prop_rows + prop_ptr[v];
    double* ob = bel_out + (size_t)v * dim * N;
    if (K == 0) { memcpy(ob, bel_in + (size_t)v * dim * N, sizeof(double) * dim * N); continue; }
    if (K == 1) { memcpy(ob, prop + (size_t)rows[0] * dim * N, sizeof(double) * dim * N); continue; }
    double* h = (double*)malloc(sizeof(double) * K * dim);
    /* dim 6 (Pose3, coordinates [t; rotation vector]): rotation matrices of every proposal point, once */
    double* Rm = dim == 6 ? (double*)malloc(sizeof(double) * 9 * (size_t)K * N) : NULL;
    int base = 0; double best = INFINITY;
    for (int l = 0; l < K; ++l) {
      const double* P = prop + (size_t)rows[l] * dim * N;
      double mean[6], sd[6];
      if (dim == 6) {
        ro_belief_spread_se3(N, P, mean, sd);
        for (int j = 0; j < N; ++j) { double w[3] = {P[3 * N + j], P[4 * N + j], P[5 * N + j]}; ro_so3_exp(w, Rm + 9 * ((size_t)l * N + j)); }
      } else if (dim == 3) ro_belief_spread_se2(N, P, P + N, P + 2 * N, mean, sd);
      else ro_belief_spread_r2(N, P, P + N, mean, sd);
      double ln = 0.0;
      for (int k = 0; k < dim; ++k) {
        h[l * dim + k] = fmax(prop_bw ? prop_bw[(size_t)rows[l] * dim + k] : cN * sd[k], 1e-6);
        ln += log(h[l * dim + k]);
      }
      if (ln < best) { best = ln; base = l; }
    }
    const int M = N;
    const double* Pb = prop + (size_t)rows[base] * dim * N;
    double* logw = (double*)malloc(sizeof(double) * M);
    double lwmax = -INFINITY;
    for (int m = 0; m < M; ++m) {
      double x[6]; for (int k = 0; k < dim; ++k) x[k] = Pb[k * N + m];
      double acc = 0.0;
      for (int l = 0; l < K; ++l) {
        if (l == base) continue;
        const double* P = prop + (size_t)rows[l] * dim * N;
        /* running log-sum-exp in particle order: sacc = Σ_j exp(-½(q_j - qmin)), qmin = smallest q so far */
        double qmin = INFINITY, sacc = 0.0;
        for (int j = 0; j < N; ++j) {
          double q = 0.0;
          if (dim == 6) {   /* tangent difference of the base point about the kernel point: (x.t - y.t, Log(R_yᵀ R_x)) */
            double U[9], wv[3];
            mat3_tmul(Rm + 9 * ((size_t)l * N + j), Rm + 9 * ((size_t)base * N + m), U); ro_so3_log(U, wv);
            for (int k = 0; k < 3; ++k) {
              double d = (x[k] - P[k * N + j]) / h[l * dim + k]; q += d * d;
              double e = wv[k] / h[l * dim + 3 + k]; q += e * e;
            }
          } else {
            for (int k = 0; k < dim; ++k) {
              double d = x[k] - P[k * N + j];
              if (dim == 3 && k == 2) d = wrap_diff(d);
              d *= 1.0 / h[l * dim + k]; q += d * d;
            }
          }
          const double dq = q - qmin;
          const double e = exp(-0.5 * fabs(dq));
          sacc = dq < 0.0 ? fma(sacc, e, 1.0) : sacc + e;
          if (q < qmin) qmin = q;
        }
        acc += -0.5 * qmin + log(sacc);
      }
      logw[m] = acc;
      if (acc > lwmax) lwmax = acc;
    }
    double* cum = (double*)malloc(sizeof(double) * M);
    double T = 0.0;
    for (int m = 0; m < M; ++m) { T += exp(logw[m] - lwmax); cum[m] = T; }
    double hp[6];
    for (int k = 0; k < dim; ++k) { double a = 0.0; for (int l = 0; l < K; ++l) a += 1.0 / (h[l * dim + k] * h[l * dim + k]); hp[k] = 1.0 / sqrt(a); }
    double u;
    { uint32_t key[2] = {(uint32_t)o->seed, (uint32_t)(o->seed >> 32)};
      uint64_t st = o->stream_offset + (uint64_t)v;
      uint32_t ctr[4] = {0xFFFFFFFFu, (uint32_t)st, (uint32_t)(st >> 32), (3u << 16)}; uint32_t w[4];
      ro_philox4x32_10(ctr, key, w);
      u = ((double)w[0] + 0.5) * (1.0 / 4294967296.0); }
    int m = 0;
    for (int i = 0; i < N; ++i) {
      const double tau = (i + u) * T / N;
      while (m < M - 1 && !(cum[m] > tau)) ++m;
      double xi[6];
      ro_rng_normals(o->seed, o->stream_offset + (uint64_t)v, (uint32_t)i, dim, xi);
      if (dim == 6) {   /* x ⊕ (h_prod ⊙ ξ): translation added, rotation R ← R Exp(h_ω ⊙ ξ_ω) */
        double e[3] = {hp[3] * xi[3], hp[4] * xi[4], hp[5] * xi[5]}, E[9], Rn[9], wn[3];
        for (int k = 0; k < 3; ++k) ob[k * N + i] = Pb[k * N + m] + hp[k] * xi[k];
        ro_so3_exp(e, E); mat3_mul(Rm + 9 * ((size_t)base * N + m), E, Rn); ro_so3_log(Rn, wn);
        for (int k = 0; k < 3; ++k) ob[(3 + k) * N + i] = wn[k];
      } else {
        for (int k = 0; k < dim; ++k) ob[k * N + i] = Pb[k * N + m] + hp[k] * xi[k];
        if (dim == 3) ob[2 * N + i] = wrap_diff(ob[2 * N + i]);
      }
    }
    free(h); free(logw); free(cum); free(Rm);
  }
  return 0;
}
int ro_product(const ro_opts* o, int dim, int V, const int32_t* prop_ptr, const int32_t* prop_rows,
               const double* prop, const double* bel_in, double* bel_out) {
  return ro_product_bw(o, dim, V, prop_ptr, prop_rows, prop, NULL, bel_in, bel_out);
}

/* ======================================================================== */
/* manifoldProduct: multiscale Gibbs sampling from a product of kernel density estimates                                       */
/* ======================================================================== */
/* ⚠AMP `manifoldProduct(ff, manifold; Niter=1)` -> ⚠KDE.jl `prodAppxMSGibbsS` (port of A. Ihler's kde toolbox): the algorithm of
 * Ihler, Sudderth, Freeman, Willsky, "Efficient multiscale sampling from products of Gaussian mixtures", NIPS 2003, restated
 * from the paper and the published implementation (neither package is vendored with RoME).
 * PARITY UNPINNED by the reference (random; no seeds): pinned statistically only -- the hexagon windows of
 * test/testHexagonal2D_CliqByCliq.jl:37-79 and the reference's solved graph examples/fg-after-solve.tar.gz (tests/).
 *
 * Every input density j (a proposal: N points, one bandwidth per coordinate) is summarised by a binary ball tree:
 *   level l has 2^l nodes, node z covers the sorted positions [floor(z N / 2^l), floor((z+1) N / 2^l)); levels 0 .. L,
 *   L = ceil(log2 N), so level L holds the single points.  Top-down, every node with >= 2 points is sorted along the coordinate
 *   of largest extent (max - min over its points; first such coordinate), which sends the lower half to its left child --
 *   Ihler's kd-tree build with index-range halving.  Node statistics, bottom-up (Chan's pairwise update, in double):
 *       n = n_l + n_r,  mean = (n_l m_l + n_r m_r)/n,  M2 = M2_l + M2_r + n_l n_r/n (m_l - m_r)²,  var = M2/n + h²
 *   i.e. the node is the moment-matched Gaussian of the kernels below it (weight n/N).  Coordinates are offsets from the
 *   density's point 0 (circular coordinates: wrapped offsets), kept in SINGLE precision like the device's tree (mean, var, 1/var
 *   and c = log(n/N) - ½ Σ log var are rounded to float once); all evaluations below are in double on those values.
 * One output sample (Philox stream = variable, counter = sample index), labels sel[j] = root:
 *   for l = 1 .. L:
 *     (a) x ~ product of the Gaussians of the currently selected nodes (level l-1)           [samplePoint]
 *     (b) every tree moves to level l                                                          [levelDown]
 *     (c) for every j: sel[j] ~ p(z) ∝ w_z N(x; m_z, var_z) over ALL nodes of level l          [sampleIndices]
 *     (d) gibbs_iters sweeps, for every j: with (M, C) the product Gaussian of the other selected nodes,
 *         sel[j] ~ p(z) ∝ w_z N(m_z; M, var_z + C) over all nodes of level l                   [sampleIndex: the Gibbs step]
 *   output = a draw from the product of the selected level-L kernels (the particles themselves, full precision).
 * Categorical draws are one-pass reservoir selections (running maximum of log p, rescaled total, accept candidate z with
 * probability a_z / T) driven by a xorshift32 stream seeded from one Philox word per draw -- the same arithmetic on both sides.
 * K = 0: the belief is kept; K = 1: the proposal is the product (AMP returns it unchanged). */
typedef struct { float key; int id; } msg_key;
static int msg_cmp(const void* a, const void* b) {
  const msg_key* p = (const msg_key*)a; const msg_key* q = (const msg_key*)b;
  if (p->key < q->key) return -1;
  if (p->key > q->key) return 1;
  return p->id - q->id;
}
typedef struct {
  int L, N, D;
  float* mean;   /* [nodes of levels 0..L-1][D]  (node (l, z) at index (1<<l) - 1 + z) */
  float* var; float* ivar; float* cz; int* cnt;
  float* ys;     /* [N][D] offsets of the points in final (sorted) order */
  int* perm;     /* sorted position -> particle index */
  float lvar[6], livar[6], lcz;   /* level-L kernels */
  double ref[6], h[6];
  double R0[9];                   /* dim 6 (Pose3): rotation of point 0 -- coordinates 3..5 are Log(R0ᵀ R_i), the chart at R0 */
} msg_tree;
static inline void msg_range(int N, int l, int z, int* a, int* b) { *a = (int)(((long long)z * N) >> l); *b = (int)(((long long)(z + 1) * N) >> l); }
static void msg_build(msg_tree* T, int N, int D, const double* P /*[D][N]*/, const double* h, uint32_t circ) {
  int L = 0; while ((1 << L) < N) ++L;
  T->L = L; T->N = N; T->D = D;
  const int nn = (1 << L) - 1 > 0 ? (1 << L) - 1 : 1;
  T->mean = (float*)calloc((size_t)nn * D, sizeof(float)); T->var = (float*)calloc((size_t)nn * D, sizeof(float));
  T->ivar = (float*)calloc((size_t)nn * D, sizeof(float)); T->cz = (float*)calloc(nn, sizeof(float)); T->cnt = (int*)calloc(nn, sizeof(int));
  T->ys = (float*)malloc(sizeof(float) * N * D); T->perm = (int*)malloc(sizeof(int) * N);
  float* y = (float*)malloc(sizeof(float) * N * D);
  for (int d = 0; d < D; ++d) {
    T->ref[d] = P[d * N]; T->h[d] = fmax(h[d], 1e-6);
    if (D == 6 && d >= 3) continue;
    for (int i = 0; i < N; ++i) {
      double o = P[d * N + i] - P[d * N];
      if ((circ >> d) & 1u) o = lcv_wrap(o);
      y[i * D + d] = (float)o + 0.0f;   /* -0 -> +0: one order for equal offsets */
    }
  }
  if (D == 6) {   /* Pose3: rotation coordinates in the chart at the rotation of point 0 */
    double w0[3] = {P[3 * N], P[4 * N], P[5 * N]};
    ro_so3_exp(w0, T->R0);
    for (int i = 0; i < N; ++i) {
      double w[3] = {P[3 * N + i], P[4 * N + i], P[5 * N + i]}, Ri[9], U[9], lg[3];
      ro_so3_exp(w, Ri); mat3_tmul(T->R0, Ri, U); ro_so3_log(U, lg);
      for (int k = 0; k < 3; ++k) y[i * D + 3 + k] = (float)lg[k] + 0.0f;
    }
  }
  for (int i = 0; i < N; ++i) T->perm[i] = i;
  msg_key* kk = (msg_key*)malloc(sizeof(msg_key) * N);
  for (int l = 0; l < L; ++l)
    for (int z = 0; z < (1 << l); ++z) {
      int a, b; msg_range(N, l, z, &a, &b);
      if (b - a < 2) continue;
      int best = 0; float ext = -1.0f;
      for (int d = 0; d < D; ++d) {
        float mn = INFINITY, mx = -INFINITY;
        for (int p = a; p < b; ++p) { const float v = y[T->perm[p] * D + d]; if (v < mn) mn = v; if (v > mx) mx = v; }
        if (mx - mn > ext) { ext = mx - mn; best = d; }
      }
      for (int p = a; p < b; ++p) { kk[p - a].key = y[T->perm[p] * D + best]; kk[p - a].id = T->perm[p]; }
      qsort(kk, (size_t)(b - a), sizeof(msg_key), msg_cmp);
      for (int p = a; p < b; ++p) T->perm[p] = kk[p - a].id;
    }
  free(kk);
  for (int p = 0; p < N; ++p) for (int d = 0; d < D; ++d) T->ys[p * D + d] = y[T->perm[p] * D + d];
  free(y);
  /* bottom-up statistics: (n, mean, M2) per node and coordinate, level L = single points */
  const int tot = (1 << (L + 1)) - 1;
  double* m = (double*)calloc((size_t)tot * D, sizeof(double)); double* M2 = (double*)calloc((size_t)tot * D, sizeof(double));
  int* n = (int*)calloc(tot, sizeof(int));
  for (int z = 0; z < (1 << L); ++z) {
    int a, b; msg_range(N, L, z, &a, &b);
    const int id = (1 << L) - 1 + z;
    n[id] = b - a;
    if (b > a) for (int d = 0; d < D; ++d) m[id * D + d] = (double)T->ys[a * D + d];
  }
  for (int l = L - 1; l >= 0; --l)
    for (int z = 0; z < (1 << l); ++z) {
      const int id = (1 << l) - 1 + z, cl = (1 << (l + 1)) - 1 + 2 * z, cr = cl + 1;
      const int nl = n[cl], nr = n[cr];
      n[id] = nl + nr;
      for (int d = 0; d < D; ++d) {
        if (nl + nr == 0) continue;
        if (nr == 0) { m[id * D + d] = m[cl * D + d]; M2[id * D + d] = M2[cl * D + d]; continue; }
        if (nl == 0) { m[id * D + d] = m[cr * D + d]; M2[id * D + d] = M2[cr * D + d]; continue; }
        const double dl = m[cl * D + d] - m[cr * D + d], nt = (double)(nl + nr);
        m[id * D + d] = ((double)nl * m[cl * D + d] + (double)nr * m[cr * D + d]) / nt;
        M2[id * D + d] = M2[cl * D + d] + M2[cr * D + d] + (double)nl * (double)nr / nt * dl * dl;
      }
      T->cnt[id] = n[id];
      if (n[id] > 0) {
        double lg = 0.0;
        for (int d = 0; d < D; ++d) {
          const double v = M2[id * D + d] / (double)n[id] + T->h[d] * T->h[d];
          T->mean[id * D + d] = (float)m[id * D + d]; T->var[id * D + d] = (float)v; T->ivar[id * D + d] = (float)(1.0 / v);
          lg += log(v);
        }
        T->cz[id] = (float)(log((double)n[id] / (double)N) - 0.5 * lg);
      }
    }
  double lg = 0.0;
  for (int d = 0; d < D; ++d) { const double v = T->h[d] * T->h[d]; T->lvar[d] = (float)v; T->livar[d] = (float)(1.0 / v); lg += log(v); }
  T->lcz = (float)(log(1.0 / (double)N) - 0.5 * lg);
  free(m); free(M2); free(n);
}
static void msg_free(msg_tree* T) { free(T->mean); free(T->var); free(T->ivar); free(T->cz); free(T->cnt); free(T->ys); free(T->perm); }
/* statistics of node z of level l (l == L: the single point at sorted position z) as doubles of the stored floats; returns the count */
static int msg_node(const msg_tree* T, int l, int z, double* mean, double* var, double* ivar, double* cz) {
  if (l < T->L) {
    const int id = (1 << l) - 1 + z;
    for (int d = 0; d < T->D; ++d) { mean[d] = (double)T->mean[id * T->D + d]; var[d] = (double)T->var[id * T->D + d]; ivar[d] = (double)T->ivar[id * T->D + d]; }
    *cz = (double)T->cz[id];
    return T->cnt[id];
  }
  const int a = z;   /* labels of level L are sorted positions */
  for (int d = 0; d < T->D; ++d) { mean[d] = (double)T->ys[a * T->D + d]; var[d] = (double)T->lvar[d]; ivar[d] = (double)T->livar[d]; }
  *cz = (double)T->lcz;
  return 1;
}
static inline uint32_t msg_xorshift(uint32_t r) { r ^= r << 13; r ^= r >> 17; r ^= r << 5; return r; }
/* Candidate arithmetic: IEEE single precision, every operation written out (this file is compiled with -ffp-contract=off; fmaf is
 * the correctly rounded fused operation).  The device evaluates two candidates per lane with packed instructions: same values
 * (its opt-in ROME_GIBBS_HWTRANS build takes msg_exp32 / msg_ln32 from the hardware transcendentals, within an ulp of these). */
#define MSG_ABSENT (-3.0e38f)   /* log p of "no candidate"; also the initial running maximum */
static inline float msg_f32_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t msg_bits_from_f32(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }
/* exp(max(x, -80)), x <= 0 (never 0: e^-80 = 1.8e-35 is below every acceptance threshold and every total); k = rint(x log2 e),
 * r = x - k ln2 in two parts, Cephes expf polynomial, 2^k added on the exponent field */
static inline float msg_exp32(float x) {
  x = fmaxf(x, -80.0f);
  const float k = rintf(x * 1.44269504f);
  float r = fmaf(k, -0.693359375f, x);
  r = fmaf(k, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float y = fmaf(p, r * r, r) + 1.0f;
  return msg_f32_from_bits(msg_bits_from_f32(y) + ((uint32_t)(int32_t)k << 23));
}
/* ln(v), v > 0 normal: v = m 2^e, ln m the degree-7 polynomial in m - 1.5 of ro_box_muller's radius */
static inline float msg_ln32(float v) {
  const uint32_t xb = msg_bits_from_f32(v);
  const float ke = (float)((int32_t)(xb >> 23) - 127);
  const float t = msg_f32_from_bits((xb & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
  float p = 0x1.4fab76p-7f;
  p = fmaf(p, t, -0x1.1d4ffp-6f);
  p = fmaf(p, t, 0x1.a972ep-6f);
  p = fmaf(p, t, -0x1.90d3ap-5f);
  p = fmaf(p, t, 0x1.94a6a8p-4f);
  p = fmaf(p, t, -0x1.c72898p-3f);
  p = fmaf(p, t, 0x1.555544p-1f);
  p = fmaf(p, t, 0x1.9f324cp-2f);
  return fmaf(ke, 0x1.62e43p-1f, p);
}
/* e - 2π rint(e / 2π), 2π = 6.28125 + 1.9353072e-3 */
static inline float msg_wrap32(float e) {
  const float k = rintf(e * 0.15915494f);
  e = fmaf(k, -6.28125f, e);
  return fmaf(k, -1.9353072e-3f, e);
}
/* one-pass categorical draw over candidate PAIRS (A, B): running maximum M of log p, total T of exp(log p - M).  ONE uniform per pair
 * (u = the top 24 bits of the next state r of a xorshift32 stream, never 0; compared as float(r | 255)·T_B against a·2^32): with T_B the
 * total including both candidates, B replaces the selection when u T_B < a_B, A when a_B <= u T_B < a_A + a_B, else the selection is
 * kept -- probabilities a_B / T_B, a_A / T_B, T_0 / T_B, the same as drawing after each candidate in turn. */
typedef struct { float M, T; uint32_t r; int sel; } msg_res;
static inline void msg_res_init(msg_res* R, uint32_t seed_word) { R->M = MSG_ABSENT; R->T = 0.0f; R->r = seed_word | 1u; R->sel = 0; }
static inline void msg_res_pair(msg_res* R, int zA, int zB, float lpA, float lpB) {
  const float Mn = fmaxf(fmaxf(R->M, lpA), lpB);
  const float T0 = R->T * msg_exp32(R->M - Mn);
  const float aA = msg_exp32(lpA - Mn), aB = msg_exp32(lpB - Mn);
  const float TA = T0 + aA, TB = TA + aB;
  R->r = msg_xorshift(R->r);
  const float lhs = (float)(R->r | 0xFFu) * TB, sAB = aA + aB;
  if (lhs < sAB * 4294967296.0f) R->sel = zA;
  if (lhs < aB * 4294967296.0f) R->sel = zB;
  R->T = TB; R->M = Mn;
}
/* Σ_d s_d / v_d over a group of gd <= 3 coordinates with one division: (Σ_d s_d Π_{e≠d} v_e) / Π_d v_d; *pv = Π_d v_d */
static inline float msg_ratio_group(int gd, const float* s, const float* v, float* pv) {
  float num, den;
  if (gd == 3) {
    const float pab = v[0] * v[1];
    num = fmaf(s[0], v[1] * v[2], fmaf(s[1], v[0] * v[2], s[2] * pab));
    den = pab * v[2];
  } else {
    num = fmaf(s[0], v[1], s[1] * v[0]);
    den = v[0] * v[1];
  }
  *pv = den;
  return num / den;
}
/* log p of candidate z of level l < L given the point e0 (in the density's chart, single precision)  [sampleIndices] */
static inline float msg_lp_point(const msg_tree* T, int l, int z, const float* e0, uint32_t circ) {
  const int id = (1 << l) - 1 + z, D = T->D;
  float q = 0.0f;
  for (int d = 0; d < D; ++d) {
    float e = e0[d] - T->mean[id * D + d];
    if ((circ >> d) & 1u) e = msg_wrap32(e);
    q = fmaf(e * e, T->ivar[id * D + d], q);
  }
  return fmaf(-0.5f, q, T->cz[id]);
}
/* log p of candidate z of level l < L given the product Gaussian (mx, cx) of the other densities  [the Gibbs step] */
static inline float msg_lp_gauss(const msg_tree* T, int l, int z, const float* mx, const float* cx, uint32_t circ) {
  const int id = (1 << l) - 1 + z, D = T->D;
  float sq[6], vv[6];
  for (int d = 0; d < D; ++d) {
    float e = T->mean[id * D + d] - mx[d];
    if ((circ >> d) & 1u) e = msg_wrap32(e);
    sq[d] = e * e;
    vv[d] = T->var[id * D + d] + cx[d];
  }
  float pv;
  float q = msg_ratio_group(D == 2 ? 2 : 3, sq, vv, &pv);
  float t = q + msg_ln32(pv);
  if (D == 6) { q = msg_ratio_group(3, sq + 3, vv + 3, &pv); t = t + (q + msg_ln32(pv)); }
  return fmaf(-0.5f, t, (float)log((double)T->cnt[id] / (double)T->N));
}
/* -½ Σ_d (y_p - c)_d² w_d of the single point at sorted position p (sign: e = ±(y - c), squared) */
static inline float msg_lp_leaf(const msg_tree* T, int p, const float* c, const float* w, int point_minus_leaf, uint32_t circ) {
  const int D = T->D;
  float q = 0.0f;
  for (int d = 0; d < D; ++d) {
    float e = point_minus_leaf ? c[d] - T->ys[p * D + d] : T->ys[p * D + d] - c[d];
    if ((circ >> d) & 1u) e = msg_wrap32(e);
    q = fmaf(e * e, w[d], q);
  }
  return q * -0.5f;
}
int ro_product_msgibbs(const ro_opts* o, int dim, int V, const int32_t* prop_ptr, const int32_t* prop_rows,
                       const double* prop /*[rows][dim][N]*/, const double* prop_bw /*[rows][dim]*/, const double* bel_in,
                       uint32_t circular_mask, int gibbs_iters, double* bel_out) {
  const int N = o->n_particles, D = dim;
  if ((dim != 2 && dim != 3 && dim != 6) || N < 1 || !prop_bw) return -1;
  if (gibbs_iters < 1) gibbs_iters = 1;
  const uint32_t key[2] = {(uint32_t)o->seed, (uint32_t)(o->seed >> 32)};
#pragma omp parallel for schedule(dynamic, 4)
  for (int v = 0; v < V; ++v) {
    const int K = prop_ptr[v + 1] - prop_ptr[v];
    const int32_t* rows = prop_rows + prop_ptr[v];
    double* ob = bel_out + (size_t)v * D * N;
    if (K == 0) { memcpy(ob, bel_in + (size_t)v * D * N, sizeof(double) * D * N); continue; }
    if (K == 1) { memcpy(ob, prop + (size_t)rows[0] * D * N, sizeof(double) * D * N); continue; }
    msg_tree* T = (msg_tree*)malloc(sizeof(msg_tree) * K);
    for (int j = 0; j < K; ++j) msg_build(&T[j], N, D, prop + (size_t)rows[j] * D * N, prop_bw + (size_t)rows[j] * D, circular_mask);
    const int L = T[0].L;
    int* sel = (int*)malloc(sizeof(int) * K);
    const uint64_t st = o->stream_offset + (uint64_t)v;
    for (int s = 0; s < N; ++s) {
      uint32_t qu = 0, qn = 0, wu[4], wn[4]; double npair[2];
      #define MSG_UNIFORM_WORD(out) do { if ((qu & 3u) == 0) { uint32_t c_[4] = {(uint32_t)s, (uint32_t)st, (uint32_t)(st >> 32), (6u << 16) | (qu >> 2)}; \
                                           ro_philox4x32_10(c_, key, wu); } (out) = wu[qu & 3u]; ++qu; } while (0)
      #define MSG_NORMAL(out) do { if ((qn & 3u) == 0) { uint32_t c_[4] = {(uint32_t)s, (uint32_t)st, (uint32_t)(st >> 32), (7u << 16) | (qn >> 2)}; \
                                     ro_philox4x32_10(c_, key, wn); } \
                                   if ((qn & 1u) == 0) ro_box_muller(wn[qn & 2u], wn[(qn & 2u) + 1], &npair[0], &npair[1]); (out) = npair[qn & 1u]; ++qn; } while (0)
      for (int j = 0; j < K; ++j) sel[j] = 0;
      double x[6];
      double xR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   /* dim 6: rotation of the current point */
      const int DE = D == 6 ? 3 : D;                /* coordinates handled one by one (Euclidean / circular) */
      for (int l = 1; l <= L + 1; ++l) {
        /* (a) a point from the product of the selected nodes of level l-1.  Deviations are taken from density 0's node
         * (circular coordinates: wrapped; rotations: Log(Q_0ᵀ Q_j) with Q_j = R0_j Exp(mean_ω) the node's absolute rotation) */
        for (int d = 0; d < DE; ++d) {
          double prec = 0.0, num = 0.0, mu0 = 0.0;
          for (int j = 0; j < K; ++j) {
            double mean[6], var[6], ivar[6], cz;
            msg_node(&T[j], l - 1, sel[j], mean, var, ivar, &cz);
            double mabs, iv;
            if (l - 1 == L) {   /* the selected kernel itself: the particle at full precision, its bandwidth in double */
              mabs = prop[(size_t)rows[j] * D * N + (size_t)d * N + T[j].perm[sel[j]]];
              iv = 1.0 / (T[j].h[d] * T[j].h[d]);
            }
            else { mabs = T[j].ref[d] + mean[d]; iv = ivar[d]; }
            if (j == 0) mu0 = mabs;
            double dev = mabs - mu0;
            if ((circular_mask >> d) & 1u) dev = lcv_wrap(dev);
            prec += iv; num += iv * dev;
          }
          double xi; MSG_NORMAL(xi);
          x[d] = mu0 + num / prec + xi / sqrt(prec);
        }
        if (D == 6) {
          double B[9], prec[3] = {0, 0, 0}, num[3] = {0, 0, 0};
          for (int j = 0; j < K; ++j) {
            double mean[6], var[6], ivar[6], cz, Q[9], iv[3];
            msg_node(&T[j], l - 1, sel[j], mean, var, ivar, &cz);
            if (l - 1 == L) {
              const int pi = T[j].perm[sel[j]];
              double w[3] = {prop[(size_t)rows[j] * D * N + 3 * (size_t)N + pi], prop[(size_t)rows[j] * D * N + 4 * (size_t)N + pi],
                             prop[(size_t)rows[j] * D * N + 5 * (size_t)N + pi]};
              ro_so3_exp(w, Q);
              for (int k = 0; k < 3; ++k) iv[k] = 1.0 / (T[j].h[3 + k] * T[j].h[3 + k]);
            } else {
              double E[9]; ro_so3_exp(mean + 3, E); mat3_mul(T[j].R0, E, Q);
              for (int k = 0; k < 3; ++k) iv[k] = ivar[3 + k];
            }
            double dev[3] = {0, 0, 0};
            if (j == 0) memcpy(B, Q, sizeof(B));
            else { double U[9]; mat3_tmul(B, Q, U); ro_so3_log(U, dev); }
            for (int k = 0; k < 3; ++k) { prec[k] += iv[k]; num[k] += iv[k] * dev[k]; }
          }
          double e[3], E[9];
          for (int k = 0; k < 3; ++k) { double xi; MSG_NORMAL(xi); e[k] = num[k] / prec[k] + xi / sqrt(prec[k]); }
          ro_so3_exp(e, E); mat3_mul(B, E, xR);
        }
        if (l == L + 1) break;   /* that was the output draw from the selected kernels */
        /* (c) labels of level l given the point (in the density's own chart -- Euclidean there -- and rounded to single) */
        for (int j = 0; j < K; ++j) {
          uint32_t w; MSG_UNIFORM_WORD(w);
          msg_res R; msg_res_init(&R, w);
          double e0d[6]; float e0[6];
          for (int d = 0; d < DE; ++d) { e0d[d] = x[d] - T[j].ref[d]; if ((circular_mask >> d) & 1u) e0d[d] = lcv_wrap(e0d[d]); }
          if (D == 6) { double U[9]; mat3_tmul(T[j].R0, xR, U); ro_so3_log(U, e0d + 3); }
          for (int d = 0; d < D; ++d) e0[d] = (float)e0d[d];
          if (l < L)
            for (int z = 0; z < (1 << l); z += 2)
              msg_res_pair(&R, z, z + 1, msg_lp_point(&T[j], l, z, e0, circular_mask), msg_lp_point(&T[j], l, z + 1, e0, circular_mask));
          else   /* single points (label = sorted position): common variance and weight -- the constants of the draw drop out */
            for (int p = 0; p < N; p += 2)
              msg_res_pair(&R, p, p + 1, msg_lp_leaf(&T[j], p, e0, T[j].livar, 1, circular_mask),
                           p + 1 < N ? msg_lp_leaf(&T[j], p + 1, e0, T[j].livar, 1, circular_mask) : MSG_ABSENT);
          sel[j] = R.sel;
        }
        /* (d) Gibbs sweeps over the labels */
        for (int it = 0; it < gibbs_iters; ++it)
          for (int j = 0; j < K; ++j) {
            double Mx[6], Cx[6];   /* product Gaussian of the other selected nodes, mean in density j's chart */
            for (int d = 0; d < DE; ++d) {
              double prec = 0.0, num = 0.0, mu0 = 0.0; int first = 1;
              for (int i = 0; i < K; ++i) {
                if (i == j) continue;
                double mean[6], var[6], ivar[6], cz;
                msg_node(&T[i], l, sel[i], mean, var, ivar, &cz);
                const double mabs = T[i].ref[d] + mean[d];
                if (first) { mu0 = mabs; first = 0; }
                double dev = mabs - mu0;
                if ((circular_mask >> d) & 1u) dev = lcv_wrap(dev);
                prec += ivar[d]; num += ivar[d] * dev;
              }
              Mx[d] = mu0 + num / prec - T[j].ref[d]; Cx[d] = 1.0 / prec;
              if ((circular_mask >> d) & 1u) Mx[d] = lcv_wrap(Mx[d]);
            }
            if (D == 6) {
              double B[9], prec[3] = {0, 0, 0}, num[3] = {0, 0, 0}; int first = 1;
              for (int i = 0; i < K; ++i) {
                if (i == j) continue;
                double mean[6], var[6], ivar[6], cz, E[9], Q[9], dev[3] = {0, 0, 0};
                msg_node(&T[i], l, sel[i], mean, var, ivar, &cz);
                ro_so3_exp(mean + 3, E); mat3_mul(T[i].R0, E, Q);
                if (first) { memcpy(B, Q, sizeof(B)); first = 0; }
                else { double U[9]; mat3_tmul(B, Q, U); ro_so3_log(U, dev); }
                for (int k = 0; k < 3; ++k) { prec[k] += ivar[3 + k]; num[k] += ivar[3 + k] * dev[k]; }
              }
              double e[3], E[9], QM[9], U[9];
              for (int k = 0; k < 3; ++k) { e[k] = num[k] / prec[k]; Cx[3 + k] = 1.0 / prec[k]; }
              ro_so3_exp(e, E); mat3_mul(B, E, QM); mat3_tmul(T[j].R0, QM, U); ro_so3_log(U, Mx + 3);
            }
            uint32_t w; MSG_UNIFORM_WORD(w);
            msg_res R; msg_res_init(&R, w);
            float mx[6], cx[6];
            for (int d = 0; d < D; ++d) { mx[d] = (float)Mx[d]; cx[d] = (float)Cx[d]; }
            if (l < L)
              for (int z = 0; z < (1 << l); z += 2)
                msg_res_pair(&R, z, z + 1, msg_lp_gauss(&T[j], l, z, mx, cx, circular_mask), msg_lp_gauss(&T[j], l, z + 1, mx, cx, circular_mask));
            else {
              float ivv[6];
              for (int d = 0; d < D; ++d) ivv[d] = 1.0f / (T[j].lvar[d] + cx[d]);
              for (int p = 0; p < N; p += 2)
                msg_res_pair(&R, p, p + 1, msg_lp_leaf(&T[j], p, mx, ivv, 0, circular_mask),
                             p + 1 < N ? msg_lp_leaf(&T[j], p + 1, mx, ivv, 0, circular_mask) : MSG_ABSENT);
            }
            sel[j] = R.sel;
          }
      }
      if (D == 6) ro_so3_log(xR, x + 3);
      for (int d = 0; d < D; ++d) ob[(size_t)d * N + s] = ((circular_mask >> d) & 1u) ? lcv_wrap(x[d]) : x[d];
      #undef MSG_UNIFORM_WORD
      #undef MSG_NORMAL
    }
    for (int j = 0; j < K; ++j) msg_free(&T[j]);
    free(T); free(sel);
  }
  return 0;
}
