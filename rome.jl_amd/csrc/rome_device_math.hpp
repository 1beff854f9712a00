// rome_device_math.hpp -- gfx950 device-side FP64 math for the factor-convolution hot path.
//
// SE(2)/SE(3) hybrid-representation group ops, the four RoME residual functors, a counter-based
// RNG and the per-particle solvers, all as per-lane register code (3x3 / 6x6 objects: no MFMA).
// Reference behaviour being reproduced (paths relative to the RoME.jl checkout):
//   src/factors/Pose2D.jl:51-67          Pose2Pose2 residual
//   src/factors/PriorPose2.jl:19-47      _vee / _compose / PriorPose2 residual
//   src/factors/BearingRange2D.jl:17-64  getSample + Pose2Point2BearingRange residual
//   src/factors/Pose3Pose3.jl:17-29      Pose3Pose3 residual
//   src/factors/Pose3D.jl:15-19          PriorPose3 residual
// plus the (unvendored) IncrementalInference `_solveLambdaNumeric` loop and Optim.jl
// Nelder-Mead defaults it runs around them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rome {

constexpr double kPi = 3.141592653589793238462643383279502884;
constexpr double kSqrtEps = 1.4901161193847656e-8;  // Julia isapprox default rtol for Float64

// ------------------------------------------------------------------------------------------
// elementary functions tuned for this path
// ------------------------------------------------------------------------------------------
// sin/cos by table and short polynomials: x = n·π/128 + r, |r| <= π/256 (two-term Cody-Waite reduction with FMA), sin/cos of
// n·π/128 from a 256-entry table of correctly rounded doubles (4 kB, one 16-byte load through the vector cache), sin r / cos r to
// r⁵ / r⁶ (truncation < 1e-17), combined by the angle-addition formulas.  16 double-precision instructions instead of the 31 + 6
// selects of the π/2 reduction with the fdlibm kernel polynomials it replaces (a double-precision instruction issues in 5.8
// cycles per wave on gfx950: 1.5 of the headline kernel's 16.7 µs were sincos).  Abs error <= 5e-16 for |x| <= 1e5 (against numpy,
// tests/test_gpu_device_math.py); every angle on this path is a pose heading, a rotation-vector norm or 2πu.
__device__ static const double kSinCosTable[512] = {
#include "rome_sincos_table.inc"
};
__device__ __forceinline__ void fast_sincos(double x, double* sn, double* cs) {
  const double n = rint(x * 40.74366543152521);                 // 128/π
  double r = fma(-n, 0x1.921fb54442d18p-6, x);                   // π/128, high and low parts
  r = fma(-n, 0x1.1a62633145c07p-60, r);
  const int k = (int)n & 255;
  const double2 t = *reinterpret_cast<const double2*>(&kSinCosTable[2 * k]);   // (sin, cos) of n·π/128
  const double z = r * r;
  const double sr = fma(r * z, fma(z, 8.33333333333333333e-03, -1.66666666666666667e-01), r);
  const double cr = fma(z, fma(z, fma(z, -1.38888888888888889e-03, 4.16666666666666667e-02), -0.5), 1.0);
  *sn = fma(t.x, cr, t.y * sr);
  *cs = fma(t.y, cr, -(t.x * sr));
}

// θ -> [-π, π]: same value as atan2(sin θ, cos θ) up to an ulp (and the ±π tie), without transcendentals.
__device__ __forceinline__ double wrap_pi(double th) {
  const double k = rint(th * 0.15915494309189533577);  // 1/(2π)
  double r = fma(-k, 6.283185307179586, th);
  r = fma(-k, 2.4492935982947064e-16, r);
  return r;
}


// 1 / v and 1 / sqrt(v) for positive normal doubles: hardware seed + two Newton steps (relative error ~1e-16; an IEEE division is
// ~25 instructions, these are 5 and 7)
__device__ __forceinline__ double fast_rcp(double v) {
  double y = __builtin_amdgcn_rcp(v);
  y = __builtin_fma(__builtin_fma(-v, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-v, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double fast_rsqrt(double v) {
  double y = __builtin_amdgcn_rsq(v);
  y = y * __builtin_fma(-0.5 * v * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * v * y, y, 1.5);
  return y;
}
// sqrt(x), x >= 0 finite: v_rsq_f64 seed + one coupled Goldschmidt step + one residual correction
// (≈ 45 SIMD-cycles per wave instead of ≈ 100 for the library call; ≤ 1 ulp on normal inputs).
__device__ __forceinline__ double fast_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  return x > 0.0 ? g : 0.0;
}
// log(u) for normal positive doubles (used on u in (0,1]): fdlibm's e_log.c kernel (< 1 ulp).
__device__ __forceinline__ double fast_log(double x) {
  const long long bits = __double_as_longlong(x);
  int hx = (int)(bits >> 32);
  const unsigned lx = (unsigned)bits;
  int k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int i = (hx + 0x95f64) & 0x100000;
  k += i >> 20;
  const double m = __longlong_as_double(((long long)(hx | (i ^ 0x3ff00000)) << 32) | lx);
  const double f = m - 1.0;
  // f / (2 + f) by v_rcp_f64 + two Newton steps on the denominator (in [1.41, 3.42]; relative error ~1e-16, the quotient then to
  // ~1.5 ulp): an IEEE double division is ~25 instructions, and the bandwidth search takes two logarithms per likelihood evaluation
  const double s = f * fast_rcp(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// exp(x) for x <= 0 (kernel-density weights): k = rint(x·log2 e), r = x − k·ln2 (two-part), degree-12 Taylor on
// |r| <= ln2/2 (truncation 1.7e-16), scaled by 2^k with v_ldexp_f64.  ≈ 20 VALU instructions, <= 2 ulp;
// arguments below −750 (including −inf) return 0.
__device__ __forceinline__ double fast_exp_neg(double x) {
  x = fmax(x, -750.0);
  const double k = rint(x * 1.4426950408889634);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 2.08767569878680989792e-09;                 // 1/12!
  p = fma(p, r, 2.50521083854417187751e-08);             // 1/11!
  p = fma(p, r, 2.75573192239858906526e-07);             // 1/10!
  p = fma(p, r, 2.75573192239858906526e-06);             // 1/9!
  p = fma(p, r, 2.48015873015873015873e-05);             // 1/8!
  p = fma(p, r, 1.98412698412698412698e-04);             // 1/7!
  p = fma(p, r, 1.38888888888888888889e-03);             // 1/6!
  p = fma(p, r, 8.33333333333333333333e-03);             // 1/5!
  p = fma(p, r, 4.16666666666666666667e-02);             // 1/4!
  p = fma(p, r, 1.66666666666666666667e-01);             // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)k);
}

// atan2(y, x) for finite arguments: one division and fdlibm's s_atan.c kernel polynomial.  a = min/max of (|x|, |y|);
// above tan(π/8) the argument is folded with atan(a) = π/4 + atan((a−1)/(a+1)), written on numerator / denominator so that
// a single division serves both cases; |b| <= tan(π/8) < 7/16, the range the polynomial is fitted on.  ≈ 45 VALU
// instructions (library call: ≈ 105); error <= 2 ulp (4.4e-16 abs over 2e6 random points against libm).
__device__ __forceinline__ double fast_atan2(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const double mx = fmax(ax, ay), mn = fmin(ax, ay);
  const bool red = mn > 0.41421356237309503 * mx;
  const double num = red ? mn - mx : mn, den = red ? mn + mx : mx;
  const double b = den > 0.0 ? num / den : 0.0;   // (reciprocal seed + Newton instead of the IEEE division: no measurable gain in the bearing-range kernels)
  const double z = b * b, w = z * z;
  const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02),
                                                  6.66107313738753120669e-02), 9.09088713343650656196e-02),
                                   1.42857142725034663711e-01), 3.33333333333329318027e-01);
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02),
                                          -7.69187620504482999495e-02), -1.11111104054623557880e-01),
                            -1.99999999998764832476e-01);
  double r = b - b * (s1 + s2);
  r = red ? 0.78539816339744830962 + r : r;
  r = ay > ax ? 1.57079632679489661923 - r : r;
  r = x < 0.0 ? kPi - r : r;
  return y < 0.0 ? -r : r;
}

// ------------------------------------------------------------------------------------------
// Philox4x32-10 (Random123).  Integer only -> bit-identical to any other conforming implementation.
// ------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one v_mad_u64_u32 per product (218 vs 319 SIMD-cycles per call for the mul_hi/mul_lo pair form)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
    // (three-input xor: one v_bitop3_b32 on gfx950 instead of two v_xor_b32)
    c = u32x4{(uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c.y, k0, 0x96), (uint32_t)p1,
              (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c.w, k1, 0x96), (uint32_t)p0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}

enum : uint32_t { kDomainNoise = 1u, kDomainEntropy = 2u };

__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {  // (0,1]
  const uint64_t x = ((uint64_t)hi << 32) | lo;
  return (double)((x >> 11) + 1) * (1.0 / 9007199254740992.0);
}

// One Box-Muller pair from two Philox words.  The transform is defined in IEEE single-precision operations (explicit fmaf
// only, contraction off) -- the same function, bit for bit, as ro_box_muller in oracle/rome_oracle.c:
//   radius  u1 = x·2^-32, x = float(wa) + 1 in [1, 2^32]: -ln u1 = (32 - e) ln2 - ln m, x = m·2^e, ln m a degree-7 polynomial
//           in m - 1.5 (|error| <= 2.7e-7; no division, no library log);
//   angle   a = (π/4)·int32(wb << 2)·2^-31 in [-π/4, π/4): the direction (cos a - sin a, cos a + sin a)/√2 is the angle π/4 + a,
//           uniform on the first quadrant; bits 31 / 30 of wb mirror it into the other three (no range reduction at all);
//   n0 = ±√(-ln u1)·(c - s), n1 = ±√(-ln u1)·(c + s).
// ≈ 50 VALU instructions per pair, none in double precision but the final conversions (the FP64 log / sqrt / sincos form it
// replaces: ≈ 135); draws carry 24-bit mantissas.
__device__ __forceinline__ void box_muller(uint32_t wa, uint32_t wb, double* n0, double* n1) {
#pragma clang fp contract(off)
  const float x = (float)wa + 1.0f;
  const uint32_t xb = __float_as_uint(x);
  const float ke = (float)(int32_t)(159u - (xb >> 23));
  const float t = __uint_as_float((xb & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
  float p = 0x1.4fab76p-7f;
  p = __builtin_fmaf(p, t, -0x1.1d4ffp-6f);
  p = __builtin_fmaf(p, t, 0x1.a972ep-6f);
  p = __builtin_fmaf(p, t, -0x1.90d3ap-5f);
  p = __builtin_fmaf(p, t, 0x1.94a6a8p-4f);
  p = __builtin_fmaf(p, t, -0x1.c72898p-3f);
  p = __builtin_fmaf(p, t, 0x1.555544p-1f);
  p = __builtin_fmaf(p, t, 0x1.9f324cp-2f);
  const float h = __builtin_fmaf(ke, 0x1.62e43p-1f, -p);
  const float rr = h > 0.0f ? __builtin_sqrtf(h) : 0.0f;   // IEEE single-precision square root (correctly rounded expansion)
  const float a = (float)(int32_t)(wb << 2) * 0x1.921fb6p-32f;
  const float z = a * a;
  float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = __builtin_fmaf(z, sp, -1.6666654611e-1f);
  const float sn = __builtin_fmaf(z * a, sp, a);
  float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = __builtin_fmaf(z, cp, 4.166664568298827e-2f);
  const float cs = __builtin_fmaf(z * z, cp, __builtin_fmaf(z, -0.5f, 1.0f));
  // single-precision products, signs by bit 31 of wb -> n0 and bit 30 -> n1, converted once: the draws are floats
  const float m0 = rr * (cs - sn), m1 = rr * (cs + sn);
  *n0 = (double)__uint_as_float(__float_as_uint(m0) ^ (wb & 0x80000000u));
  *n1 = (double)__uint_as_float(__float_as_uint(m1) ^ ((wb << 1) & 0x80000000u));
}
__device__ __forceinline__ u32x4 noise_words(uint64_t seed, uint64_t stream, uint32_t particle, uint32_t b) {
  return philox4x32_10(u32x4{particle, (uint32_t)stream, (uint32_t)(stream >> 32), (kDomainNoise << 16) | b},
                       (uint32_t)seed, (uint32_t)(seed >> 32));
}

// D standard normals for (seed, stream, particle): Box-Muller on Philox words.  NEIGHBOURING particles 2j and 2j + 1 -- the two
// particles one thread owns in the convolution kernels (adjacent in every SoA belief row: one 16-byte access per coordinate) --
// draw from the SAME Philox calls, made with the EVEN particle id as counter (the rule depends on the particle id only, not on N or
// the launch shape).  Oracle: ro_rng_normals.
//   D = 2, 6 (bearing-range, Pose3):  the 4·(D/2) words of calls b = 0 .. D/2 - 1 in order; the even particle takes words [0, D), the
//                                     odd one words [D, 2D); consecutive word pairs (radius word, angle word) -> one Box-Muller pair
//   D = 3 (Pose2 measurements):       ONE call for the six normals of the two particles: its 128 bits are cut into six 21-bit fields
//        field 0 / 1 = top 21 bits of x / y   -> pair A -> normals 0, 1 of the even particle
//        field 2 / 3 = top 21 bits of z / w   -> pair B -> normals 0, 1 of the odd particle
//        field 4     = low 11 bits of x, then bits 10..1 of y ; field 5 = the same of z, w
//                                              -> pair C -> normal 2 of the even particle (first output) and of the odd one (second)
//        a field f enters box_muller as the word (f << 11) | 0x400, i.e. the centre of its bin: the radius is
//        sqrt(-2 ln((f + 0.5) / 2^21)) <= 5.52, the angle has 19 bits below the two quadrant bits.  The draws differ from N(0,1)
//        by < 2e-6 in Kolmogorov distance (tests/test_host_logic.py); a Pose2 convolution of N = 100 particles costs 50 Philox calls.
template <int D>
__device__ __forceinline__ void normals3_fields(const u32x4& w, uint32_t (&r)[3], uint32_t (&a)[3]) {
  r[0] = (w.x & 0xFFFFF800u) | 0x400u; a[0] = (w.y & 0xFFFFF800u) | 0x400u;
  r[1] = (w.z & 0xFFFFF800u) | 0x400u; a[1] = (w.w & 0xFFFFF800u) | 0x400u;
  r[2] = (w.x << 21) | ((w.y & 0x7FEu) << 10) | 0x400u;
  a[2] = (w.z << 21) | ((w.w & 0x7FEu) << 10) | 0x400u;
}
// the normals of particles p_even (even) and p_even + 1 together
template <int D>
__device__ __forceinline__ void rng_normals_pair(uint64_t seed, uint64_t stream, uint32_t p_even, double (&oe)[D], double (&oo)[D]) {
  static_assert(D == 2 || D == 3 || D == 6, "measurement dimensions of the supported factors");
  if constexpr (D == 3) {
    const u32x4 w = noise_words(seed, stream, p_even, 0u);
    uint32_t r[3], a[3];
    normals3_fields<3>(w, r, a);
    box_muller(r[0], a[0], &oe[0], &oe[1]);
    box_muller(r[1], a[1], &oo[0], &oo[1]);
    box_muller(r[2], a[2], &oe[2], &oo[2]);
  } else {
    uint32_t ww[2 * D];
#pragma unroll
    for (int b = 0; b < D / 2; ++b) {
      const u32x4 w = noise_words(seed, stream, p_even, (uint32_t)b);
      ww[4 * b] = w.x; ww[4 * b + 1] = w.y; ww[4 * b + 2] = w.z; ww[4 * b + 3] = w.w;
    }
#pragma unroll
    for (int k = 0; k < D; k += 2) {
      box_muller(ww[k], ww[k + 1], &oe[k], &oe[k + 1]);
      box_muller(ww[D + k], ww[D + k + 1], &oo[k], &oo[k + 1]);
    }
  }
}
// one particle on its own (prior sampling, one-particle-per-lane launches): only the calls / pairs it needs
template <int D>
__device__ __forceinline__ void rng_normals(uint64_t seed, uint64_t stream, uint32_t particle, double (&out)[D]) {
  static_assert(D == 2 || D == 3 || D == 6, "measurement dimensions of the supported factors");
  const uint32_t pe = particle & ~1u;
  const bool odd = (particle & 1u) != 0u;
  if constexpr (D == 3) {
    const u32x4 w = noise_words(seed, stream, pe, 0u);
    uint32_t r[3], a[3];
    normals3_fields<3>(w, r, a);
    box_muller(odd ? r[1] : r[0], odd ? a[1] : a[0], &out[0], &out[1]);
    double c, s;
    box_muller(r[2], a[2], &c, &s);
    out[2] = odd ? s : c;
  } else if constexpr (D == 2) {
    const u32x4 w = noise_words(seed, stream, pe, 0u);
    box_muller(odd ? w.z : w.x, odd ? w.w : w.y, &out[0], &out[1]);
  } else {   // D == 6: even -> call 0 (x,y | z,w) + call 1 (x,y); odd -> call 1 (z,w) + call 2 (x,y | z,w)
    const u32x4 w1 = noise_words(seed, stream, pe, 1u);
    const u32x4 w02 = noise_words(seed, stream, pe, odd ? 2u : 0u);
    if (odd) {
      box_muller(w1.z, w1.w, &out[0], &out[1]); box_muller(w02.x, w02.y, &out[2], &out[3]); box_muller(w02.z, w02.w, &out[4], &out[5]);
    } else {
      box_muller(w02.x, w02.y, &out[0], &out[1]); box_muller(w02.z, w02.w, &out[2], &out[3]); box_muller(w1.x, w1.y, &out[4], &out[5]);
    }
  }
}

// The entropy uniforms as the oracle defines them (ro_rng_entropy): one Philox call per particle and cycle (two for D = 6), one
// 32-bit word per coordinate.  Used wherever the jitter can reach the proposal: Nelder-Mead, the bearing-range pose direction.
template <int D>
__device__ __forceinline__ void rng_entropy_exact(uint64_t seed, uint64_t stream, uint32_t particle, int cycle, double (&out)[D]) {
#pragma unroll
  for (int b = 0; 3 * b < D; ++b) {
    const u32x4 w = philox4x32_10(u32x4{particle, (uint32_t)stream, (uint32_t)(stream >> 32), (kDomainEntropy << 16) | (uint32_t)(2 * cycle + b)},
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t ww[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int k = 3 * b; k < D && k < 3 * b + 3; ++k) out[k] = ((double)ww[k - 3 * b] + 0.5) * (1.0 / 4294967296.0);
  }
}

// ------------------------------------------------------------------------------------------
// wave64 reductions (all lanes end with the same bits: xor-butterfly of commutative adds)
// ------------------------------------------------------------------------------------------
// Row (16-lane) butterflies are DPP moves (quad_perm / row_half_mirror / row_mirror); the four row sums
// are then folded with row_bcast:15 / row_bcast:31 and read back from lane 63.  ~4x cheaper than ds_bpermute shuffles.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // full-wave permutations within rows: every lane receives a value, so the destination needs no prior contents
  // (mov_dpp = update_dpp with an undefined `old`: saves the two zero-initialising v_mov + s_nop per 64-bit value and step)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_masked(double v) {   // rows outside ROW_MASK receive 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// gfx950 lane-group swaps: v + (the same lane of the neighbouring 16-lane row) / (of the other 32-lane half).  v_permlane16_swap
// exchanges the odd rows of its first operand with the even rows of its second: fed two copies of v it leaves (r0, r0, r2, r2) and
// (r1, r1, r3, r3), whose sum is the all-reduce of the row pairs; v_permlane32_swap does the same for the two halves of the wave.
__device__ __forceinline__ double rowpair_allsum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double halves_allsum(double v) {
  const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// FOUR sums at once, transposed: the first two butterfly steps HALVE the number of values a lane carries (even lanes keep v0, v1
// and hand v2, v3 to their neighbour, then lanes 0/1 mod 4 keep one value each and hand the other over), the remaining four steps
// run on ONE value per lane with partners that have the same lane mod 4 (row rotations by 4 and 8, then the row-pair / half swaps).
// 37 VALU instructions instead of 4 x (6 steps x 3) = 72 + 16 for the masked steps; the total of v[k] ends in every lane whose
// (lane & 3) is (0, 2, 1, 3)[k] and is read back through an SGPR pair.  (All convolution kernels take this function: identical bits.)
__device__ __forceinline__ void wave_sum4(double (&v)[4]) {
  const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const bool b0 = lane & 1u, b1 = lane & 2u;
  const double kA = b0 ? v[2] : v[0], kB = b0 ? v[3] : v[1], sA = b0 ? v[0] : v[2], sB = b0 ? v[1] : v[3];
  const double wA = kA + dpp_mov<0xB1>(sA), wB = kB + dpp_mov<0xB1>(sB);      // quad_perm [1,0,3,2]: partner lane ^ 1
  const double kC = b1 ? wB : wA, sC = b1 ? wA : wB;
  double x = kC + dpp_mov<0x4E>(sC);                                           // quad_perm [2,3,0,1]: partner lane ^ 2
  x += dpp_mov<0x124>(x);                                                      // row_ror:4
  x += dpp_mov<0x128>(x);                                                      // row_ror:8: every lane of a row holds the row sum of ITS value
  x = rowpair_allsum(x);
  x = halves_allsum(x);
  v[0] = readlane_f64(x, 0); v[1] = readlane_f64(x, 2); v[2] = readlane_f64(x, 1); v[3] = readlane_f64(x, 3);
}
template <int K>
__device__ __forceinline__ void wave_sum_n(double (&v)[K]) {
  if constexpr (K == 4) { wave_sum4(v); return; }
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov<0xB1>(v[k]);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov<0x4E>(v[k]);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov<0x141>(v[k]);  // row_half_mirror
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov<0x140>(v[k]);  // row_mirror
  // every lane of a row now holds its row sum: fold the rows with the GFX9 row broadcasts (lane 15 of rows 0/2 -> rows 1/3,
  // then lane 31 -> rows 2/3), the total lands in row 3 and is broadcast from lane 63 through an SGPR pair.
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_masked<0x142, 0xA>(v[k]);  // row_bcast:15 row_mask:0xa
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_masked<0x143, 0xC>(v[k]);  // row_bcast:31 row_mask:0xc
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = readlane_f64(v[k], 63);
}
// single-precision variant: the DPP moves fold into v_add_f32_dpp (one instruction per step instead of two moves + an add)
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_masked_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int K>
__device__ __forceinline__ void wave_sum_n_f32(float (&v)[K]) {
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_f32<0xB1>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_f32<0x4E>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_f32<0x141>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_f32<0x140>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_masked_f32<0x142, 0xA>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] += dpp_mov_masked_f32<0x143, 0xC>(v[k]);
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[k]), 63));
}
__device__ __forceinline__ double wave_sum(double v) { double a[1] = {v}; wave_sum_n<1>(a); return a[0]; }

// ------------------------------------------------------------------------------------------
// SE(2)
// ------------------------------------------------------------------------------------------
struct Se2 { double x, y, c, s; };  // point ((x,y), R=[c -s; s c])

__device__ __forceinline__ Se2 se2_from_coords(double x, double y, double th) {
  Se2 p; p.x = x; p.y = y; fast_sincos(th, &p.s, &p.c); return p;
}
// Manifolds.sym_rem: (x ≈ π ? -π : rem(x, 2π, RoundNearest)).  The IEEE remainder is evaluated as x − k·2π with k = rint(x/2π) and
// 2π split in two parts (wrap_pi): exact for |x| < 2π·2^20 up to 2.5e-16·|k|, i.e. the correctly rounded result to an ulp of π for
// every angle on this path -- instead of ocml's remainder() (a division-free but ~100-instruction exponent loop).
__device__ __forceinline__ double sym_rem(double x) {
  const double m = fabs(x) > kPi ? fabs(x) : kPi;
  const double r = wrap_pi(x);
  return fabs(x - kPi) <= kSqrtEps * m ? -kPi : r;
}

// Pose2Pose2: r = vee(log(q, p ∘ exp_ϵ(X)));  X = ((zx,zy), skew(zθ)) with (cz,sz)=cos/sin(zθ)
__device__ __forceinline__ void residual_pose2pose2(double zx, double zy, double cz, double sz,
                                                    const Se2& p, const Se2& q, double (&r)[3]) {
  const double qhx = p.x + p.c * zx - p.s * zy;
  const double qhy = p.y + p.s * zx + p.c * zy;
  const double h11 = p.c * cz - p.s * sz;
  const double h21 = p.s * cz + p.c * sz;
  const double U11 = q.c * h11 + q.s * h21;
  const double U21 = q.c * h21 - q.s * h11;
  r[0] = qhx - q.x; r[1] = qhy - q.y; r[2] = fast_atan2(U21, U11);   // (<= 2 ulp; the reference's KATs hold at 1e-14)
}
// PriorPose2: r = vee(log(p, m))
__device__ __forceinline__ void residual_priorpose2(const Se2& m, const Se2& p, double (&r)[3]) {
  const double U11 = p.c * m.c + p.s * m.s;
  const double U21 = p.c * m.s - p.s * m.c;
  r[0] = m.x - p.x; r[1] = m.y - p.y; r[2] = fast_atan2(U21, U11);
}
// Pose2Point2BearingRange: pl = p.Rᵀ (l - p.t);  r = (sym_rem(b - atan2(pl)), ρ - ‖pl‖)
__device__ __forceinline__ void residual_bearingrange(double b, double rho, const Se2& p, double lx, double ly,
                                                      double (&r)[2]) {
  const double dx = lx - p.x, dy = ly - p.y;
  const double plx = p.c * dx + p.s * dy;
  const double ply = p.c * dy - p.s * dx;
  r[0] = sym_rem(b - fast_atan2(ply, plx));
  r[1] = rho - fast_sqrt(plx * plx + ply * ply);
}

// entropy: u ← u ∘ exp_ϵ(hat(e))
__device__ __forceinline__ void se2_add_entropy(double (&t)[3], double spread, const double (&u)[3]) {
  const double ex = spread * (u[0] - 0.5), ey = spread * (u[1] - 0.5), et = spread * (u[2] - 0.5);
  double s, c; fast_sincos(t[2], &s, &c);
  t[0] += c * ex - s * ey; t[1] += s * ex + c * ey; t[2] += et;
}

// ------------------------------------------------------------------------------------------
// SO(3)/SE(3) (column-major 3x3)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) C[i + 3 * j] = A[i] * B[3 * j] + A[i + 3] * B[1 + 3 * j] + A[i + 6] * B[2 + 3 * j];
}
__device__ __forceinline__ void mat3_tmul(const double* A, const double* B, double* C) {  // Aᵀ B
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      C[i + 3 * j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[1 + 3 * j] + A[3 * i + 2] * B[2 + 3 * j];
}
__device__ __forceinline__ void mat3_vec(const double* A, const double* v, double* o) {
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[i + 3] * v[1] + A[i + 6] * v[2];
}

// Rodrigues as Manifolds' exp!(::Rotations{3}): a = sinθ/θ, b = (1-cosθ)/θ², I + aX + bX²
__device__ __forceinline__ void so3_exp(const double* w, double* R) {
  const double x = w[0], y = w[1], z = w[2];
  const double th2 = x * x + y * y + z * z;
  const double th = sqrt(th2);
  double a = 1.0, b = 0.0;
  if (th != 0.0) { double s, c; fast_sincos(th, &s, &c); a = s / th; b = (1.0 - c) / th2; }
  R[0] = 1.0 + b * (x * x - th2); R[3] = -a * z + b * x * y;      R[6] = a * y + b * x * z;
  R[1] = a * z + b * x * y;       R[4] = 1.0 + b * (y * y - th2); R[7] = -a * x + b * y * z;
  R[2] = -a * y + b * x * z;      R[5] = a * x + b * y * z;       R[8] = 1.0 + b * (z * z - th2);
}
// Manifolds' log!(::Rotations{3}) incl. the cosθ ≈ -1 branch
__device__ __forceinline__ void so3_log(const double* U, double* w) {
  const double c = 0.5 * (U[0] + U[4] + U[8] - 1.0);
  const double sx = U[5] - U[7], sy = U[6] - U[2], sz = U[1] - U[3];
  if (fabs(c + 1.0) <= kSqrtEps) {
    const double d0 = 0.5 * (U[0] + 1.0), d1 = 0.5 * (U[4] + 1.0), d2 = 0.5 * (U[8] + 1.0);
    double ax, ay, az;
    if (d0 >= d1 && d0 >= d2) { ax = d0; ay = 0.25 * (U[1] + U[3]); az = 0.25 * (U[2] + U[6]); }
    else if (d1 >= d2)        { ax = 0.25 * (U[1] + U[3]); ay = d1; az = 0.25 * (U[5] + U[7]); }
    else                      { ax = 0.25 * (U[2] + U[6]); ay = 0.25 * (U[5] + U[7]); az = d2; }
    const double n = sqrt(ax * ax + ay * ay + az * az);
    const double sgn = (ax * sx + ay * sy + az * sz) < 0.0 ? -1.0 : 1.0;
    const double k = sgn * kPi / n;
    w[0] = k * ax; w[1] = k * ay; w[2] = k * az;
    return;
  }
  double usinc;
  if (c >= 1.0) usinc = 1.0;
  else if (c <= -1.0) usinc = 0.0;
  else usinc = sqrt(1.0 - c * c) / acos(c);
  const double k = 0.5 / usinc;
  w[0] = k * sx; w[1] = k * sy; w[2] = k * sz;
}

struct Se3 { double t[3]; double R[9]; };
__device__ __forceinline__ void se3_from_coords(const double* c, Se3& p) {
  p.t[0] = c[0]; p.t[1] = c[1]; p.t[2] = c[2]; so3_exp(c + 3, p.R);
}
__device__ __forceinline__ void se3_to_coords(const Se3& p, double* c) {
  c[0] = p.t[0]; c[1] = p.t[1]; c[2] = p.t[2]; so3_log(p.R, c + 3);
}
// Pose3Pose3: r = coords(log(q, p ∘ exp_ϵ(X)));  zt = X translation, Z = Exp(z_ω)
__device__ __forceinline__ void residual_pose3pose3(const double* zt, const double* Z, const Se3& p, const Se3& q,
                                                    double (&r)[6]) {
  double Rh[9], U[9], v[3];
  mat3_mul(p.R, Z, Rh);
  mat3_vec(p.R, zt, v);
  mat3_tmul(q.R, Rh, U);
  r[0] = p.t[0] + v[0] - q.t[0]; r[1] = p.t[1] + v[1] - q.t[1]; r[2] = p.t[2] + v[2] - q.t[2];
  so3_log(U, &r[3]);
}
__device__ __forceinline__ void residual_priorpose3(const Se3& m, const Se3& p, double (&r)[6]) {
  double U[9];
  mat3_tmul(p.R, m.R, U);
  r[0] = m.t[0] - p.t[0]; r[1] = m.t[1] - p.t[1]; r[2] = m.t[2] - p.t[2];
  so3_log(U, &r[3]);
}
__device__ __forceinline__ void se3_add_entropy(Se3& T, double spread, const double (&u)[6]) {
  double e[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) e[k] = spread * (u[k] - 0.5);
  double E[9], Rn[9], v[3];
  mat3_vec(T.R, e, v);
  T.t[0] += v[0]; T.t[1] += v[1]; T.t[2] += v[2];
  so3_exp(e + 3, E); mat3_mul(T.R, E, Rn);
#pragma unroll
  for (int k = 0; k < 9; ++k) T.R[k] = Rn[k];
}

// ------------------------------------------------------------------------------------------
// Unit quaternions (w, x, y, z) -- the rotation state of the SE(3) convolution kernel.  Same group elements as the
// 3x3 frames above (Exp/Log agree with so3_exp/so3_log to rounding, and are better conditioned at θ -> π), at 4
// instead of 9 doubles per rotation and 16 instead of 27 multiply-adds per composition.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void quat_exp(const double* w, double (&q)[4]) {   // Exp(ω): (cos θ/2, sin(θ/2)/θ · ω)
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  // 1/θ by v_rsq_f64 + two Newton steps: θ = θ²·(1/θ) and sin(θ/2)/θ = sin(θ/2)·(1/θ) without the square root's correction steps and
  // without an IEEE division (~25 instructions; round 6: three Exp per particle of the Pose3Pose3 sweeps)
  const bool big = th2 > 1e-16;
  const double ith = fast_rsqrt(big ? th2 : 1.0);
  const double th = th2 * ith;
  double s, c;
  fast_sincos(0.5 * th, &s, &c);
  const double k = big ? s * ith : 0.5 - th2 * (1.0 / 48.0);
  q[0] = c; q[1] = k * w[0]; q[2] = k * w[1]; q[3] = k * w[2];
}
// Rotation angle θ in [0, π] of a unit quaternion with |vector part| = n and |w| = aw (n² + aw² = 1):
// θ = 2·asin(n) = π − 2·asin(aw), always evaluated on the branch whose argument is <= 1/√2 (well conditioned;
// measured 5 % faster in the SE(3) Newton kernel than 2·fast_atan2(n, aw)).
__device__ __forceinline__ double quat_angle(double n, double aw) {
  const double a = asin(fmin(n, aw));
  return n <= aw ? 2.0 * a : kPi - 2.0 * a;
}
// Log(q) as a rotation vector with θ in [0, π] (q and −q are the same rotation: the w >= 0 representative is used).
// Like Manifolds' log!(::Rotations{3}) (so3_log above), rotations with cos θ + 1 <= √eps are returned as θ = π exactly.
__device__ __forceinline__ void quat_log(const double (&q)[4], double* w) {
  const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const bool big = n2 > 1e-16;
  const double in = fast_rsqrt(big ? n2 : 1.0);   // 1/n (round 6: no square-root correction steps, no IEEE divisions)
  const double n = n2 * in;
  const double aw = fabs(q[0]);
  double k = big ? quat_angle(n, aw) * in : 2.0 * fast_rcp(fmax(aw, 1e-300));
  if (2.0 * aw * aw <= kSqrtEps) k = kPi * in;
  k = q[0] < 0.0 ? -k : k;
  w[0] = k * q[1]; w[1] = k * q[2]; w[2] = k * q[3];
}
__device__ __forceinline__ void quat_mul(const double (&a)[4], const double (&b)[4], double (&o)[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
__device__ __forceinline__ void quat_cmul(const double (&a)[4], const double (&b)[4], double (&o)[4]) {   // conj(a) ⊗ b
  o[0] = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  o[1] = a[0] * b[1] - a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
  o[2] = a[0] * b[2] + a[1] * b[3] - a[2] * b[0] - a[3] * b[1];
  o[3] = a[0] * b[3] - a[1] * b[2] + a[2] * b[1] - a[3] * b[0];
}
__device__ __forceinline__ void quat_mulc(const double (&a)[4], const double (&b)[4], double (&o)[4]) {   // a ⊗ conj(b)
  o[0] = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  o[1] = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
  o[2] = -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1];
  o[3] = -a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
}
__device__ __forceinline__ void quat_rot(const double (&q)[4], const double* v, double (&o)[3]) {   // R(q) v
  const double tx = 2.0 * (q[2] * v[2] - q[3] * v[1]), ty = 2.0 * (q[3] * v[0] - q[1] * v[2]), tz = 2.0 * (q[1] * v[1] - q[2] * v[0]);
  o[0] = v[0] + q[0] * tx + (q[2] * tz - q[3] * ty);
  o[1] = v[1] + q[0] * ty + (q[3] * tx - q[1] * tz);
  o[2] = v[2] + q[0] * tz + (q[1] * ty - q[2] * tx);
}

// ------------------------------------------------------------------------------------------
// Nelder-Mead with Optim.jl's defaults (AdaptiveParameters, AffineSimplexer(0.025, 0.5), g_tol test
// on sqrt(var(f)·n/(n+1)), best-vertex-or-centroid result).  The simplex is kept SORTED in fixed
// register slots (compile-time indices only -> no scratch); ties are broken by the vertex's storage
// slot exactly like sortperm's stable order.
// ------------------------------------------------------------------------------------------
template <int n>
__device__ __forceinline__ void nm_cswap(double& fa, int& sa, double (&xa)[n], double& fb, int& sb, double (&xb)[n]) {  // ensure a <= b
  const bool sw = (fb < fa) || (fb == fa && sb < sa);
  const double f0 = sw ? fb : fa, f1 = sw ? fa : fb;
  const int s0 = sw ? sb : sa, s1 = sw ? sa : sb;
  fa = f0; fb = f1; sa = s0; sb = s1;
#pragma unroll
  for (int k = 0; k < n; ++k) { const double x0 = sw ? xb[k] : xa[k], x1 = sw ? xa[k] : xb[k]; xa[k] = x0; xb[k] = x1; }
}

template <int n, class Cost>
__device__ __forceinline__ int nelder_mead(const Cost& cost, double (&x)[n], int max_iters, double g_tol) {
  constexpr int m = n + 1;
  const double alpha = 1.0, beta = 1.0 + 2.0 / n, gamma = 0.75 - 1.0 / (2.0 * n), delta = 1.0 - 1.0 / n;
  double F[m], X[m][n];   // plain arrays with compile-time indices only: scalarised into registers
  int SL[m];
#pragma unroll
  for (int i = 0; i < m; ++i) {
#pragma unroll
    for (int k = 0; k < n; ++k) X[i][k] = x[k];
    if (i > 0) X[i][i - 1] = (1.0 + 0.5) * X[i][i - 1] + 0.025;
    SL[i] = i;
    F[i] = cost(X[i]);
  }
  // full sort (insertion network)
#pragma unroll
  for (int i = 1; i < m; ++i)
#pragma unroll
    for (int j = i; j > 0; --j) nm_cswap<n>(F[j - 1], SL[j - 1], X[j - 1], F[j], SL[j], X[j]);

  // Optim's stopping statistic is sqrt(var(f)·n/(n+1)) <= g_tol with the corrected variance; evaluated here as
  // Σ(f-mean)² <= g_tol²·(n+1)  (same decision, no FP64 divide / sqrt in the loop: ≈ 250 SIMD-cycles per iteration)
  const double thr2 = g_tol * g_tol * (double)m;
  bool converged;
  {
    double mean = 0.0, v = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) mean += F[i];
    mean *= (1.0 / m);
#pragma unroll
    for (int i = 0; i < m; ++i) { const double d = F[i] - mean; v += d * d; }
    converged = v <= thr2;
  }
  int iter = 0;
  while (!converged && iter < max_iters) {
    ++iter;
    double xc[n], xr[n], xt[n];
#pragma unroll
    for (int k = 0; k < n; ++k) {   // centroid of all but the highest
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < m - 1; ++i) s += X[i][k];
      xc[k] = s * (1.0 / n);
    }
    const double f_lowest = F[0], f_second = F[m - 2], f_highest = F[m - 1];
#pragma unroll
    for (int k = 0; k < n; ++k) xr[k] = xc[k] + alpha * (xc[k] - X[m - 1][k]);
    const double f_reflect = cost(xr);
    // One second trial point per iteration with a per-lane coefficient (expansion β, outside contraction γ,
    // inside contraction -γ; unused when the reflection is simply accepted): the four Optim branches become
    // selects, so lanes of a wave taking different branches do not serialise two extra cost evaluations.
    const bool expand = f_reflect < f_lowest;
    const bool accept_reflect = !expand && f_reflect < f_second;
    const bool outside = f_reflect < f_highest;
    const double kcoef = expand ? beta : (outside ? gamma : -gamma);
#pragma unroll
    for (int k = 0; k < n; ++k) xt[k] = xc[k] + kcoef * (xr[k] - xc[k]);
    const double f_trial = cost(xt);
    const bool take_trial = expand ? (f_trial < f_reflect) : (!accept_reflect && f_trial < (outside ? f_reflect : f_highest));
    const bool take_reflect = (expand && !take_trial) || accept_reflect;
    const bool shrink = !take_trial && !take_reflect;
    if (!shrink) {
      F[m - 1] = take_trial ? f_trial : f_reflect;
#pragma unroll
      for (int k = 0; k < n; ++k) X[m - 1][k] = take_trial ? xt[k] : xr[k];
    }
    if (shrink) {
#pragma unroll
      for (int i = 1; i < m; ++i) {
#pragma unroll
        for (int k = 0; k < n; ++k) X[i][k] = X[0][k] + delta * (X[i][k] - X[0][k]);
        F[i] = cost(X[i]);
      }
#pragma unroll
      for (int i = 1; i < m; ++i)
#pragma unroll
        for (int j = i; j > 0; --j) nm_cswap<n>(F[j - 1], SL[j - 1], X[j - 1], F[j], SL[j], X[j]);
    } else {
      // only the last vertex changed: bubble it into place
#pragma unroll
      for (int j = m - 1; j > 0; --j) nm_cswap<n>(F[j - 1], SL[j - 1], X[j - 1], F[j], SL[j], X[j]);
    }
    double mean = 0.0, v = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) mean += F[i];
    mean *= (1.0 / m);
#pragma unroll
    for (int i = 0; i < m; ++i) { const double d = F[i] - mean; v += d * d; }
    converged = v <= thr2;
  }
  double xc[n];
#pragma unroll
  for (int k = 0; k < n; ++k) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < m - 1; ++i) s += X[i][k];
    xc[k] = s * (1.0 / n);
  }
  const double fcen = cost(xc);
  const bool usec = fcen < F[0];
#pragma unroll
  for (int k = 0; k < n; ++k) x[k] = usec ? xc[k] : X[0][k];
  return converged ? 0 : 1;
}

}  // namespace rome
