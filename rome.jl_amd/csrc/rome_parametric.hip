// rome_parametric.hip -- batched whitened residuals + analytic Jacobians of the hot-path factors
// (SURVEY.md §8(f) row 3 / BASELINE.json configs[4]: "parametric Gauss-Newton batched Jacobians").
//
// Replaces the per-factor residual + AD/finite-difference Jacobian evaluation IIF.solveGraphParametric!
// performs through the RoME functors evaluated at the measurement mean (SURVEY §3.4;
// getMeasurementParametric, src/factors/BearingRange2D.jl:30-37):
//     cost = Σ_f ‖ W_f r_f(μ_f ; x) ‖² ,   W_fᵀ W_f = Σ_f⁻¹
// Tangent / perturbation convention = the reference's hybrid (product-manifold) representation:
//     Pose2  x ⊕ δ = ((t + δ_t), R(θ + δ_θ))            δ = (δx, δy, δθ)
//     Point2 l ⊕ δ = l + δ
//     Pose3  x ⊕ δ = ((t + δ_t), R·Exp(δ_ω))            δ = (δt(3), δω(3))
// One thread per factor row, rows staged through LDS in blocks of 64 (k_lin below); rows are AoS coordinates.  FP64.
#include "../../include/rome_mi355.h"
#include "rome_device_math.hpp"
#include "rome_kernels.h"

namespace rome {

template <int DR, int DV>
__device__ __forceinline__ void store_whitened(const double* W, const double (&J)[DR][DV], double* out) {
  // out[DR][DV] = W[DR][DR] * J
#pragma unroll
  for (int i = 0; i < DR; ++i)
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < DR; ++k) s += W[i * DR + k] * J[k][j];
      out[i * DV + j] = s;
    }
}
template <int DR>
__device__ __forceinline__ void store_whitened_vec(const double* W, const double (&r)[DR], double* out) {
#pragma unroll
  for (int i = 0; i < DR; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < DR; ++k) s += W[i * DR + k] * r[k];
    out[i] = s;
  }
}

// inverse right / left Jacobians of SO(3):  J_r⁻¹(φ) = I + ½[φ]× + c[φ]×² ,  J_l⁻¹(φ) = I - ½[φ]× + c[φ]×²
__device__ __forceinline__ void so3_jinv(const double* phi, double sign_half, double (&J)[3][3]) {
  const double x = phi[0], y = phi[1], z = phi[2];
  const double th2 = x * x + y * y + z * z;
  double c;
  if (th2 < 1e-8) c = 1.0 / 12.0 + th2 / 720.0;
  else {
    const double th = fast_sqrt(th2);
    double s, co; fast_sincos(th, &s, &co);
    c = 1.0 / th2 - (1.0 + co) / (2.0 * th * s);
  }
  const double h = 0.5 * sign_half;
  // [φ]× = [0 -z y; z 0 -x; -y x 0] ; [φ]×² = φφᵀ - θ² I
  J[0][0] = 1.0 + c * (x * x - th2); J[0][1] = -h * z + c * x * y;       J[0][2] = h * y + c * x * z;
  J[1][0] = h * z + c * x * y;       J[1][1] = 1.0 + c * (y * y - th2);  J[1][2] = -h * x + c * y * z;
  J[2][0] = -h * y + c * x * z;      J[2][1] = h * x + c * y * z;        J[2][2] = 1.0 + c * (z * z - th2);
}

__device__ __forceinline__ void lin_row_priorpose2(const double* mu, const double* W, const double* xa, double* r, double* Ja) {
  const Se2 M = se2_from_coords(mu[0], mu[1], mu[2]);
  const Se2 P = se2_from_coords(xa[0], xa[1], xa[2]);
  double rr[3]; residual_priorpose2(M, P, rr);
  const double J[3][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
  store_whitened_vec<3>(W, rr, r);
  store_whitened<3, 3>(W, J, Ja);
}
__device__ __forceinline__ void lin_row_priorpoint2(const double* mu, const double* W, const double* xa, double* r, double* Ja) {
  const double rr[2] = {mu[0] - xa[0], mu[1] - xa[1]};  // src/factors/Point2D.jl:14-18
  const double J[2][2] = {{-1, 0}, {0, -1}};
  store_whitened_vec<2>(W, rr, r);
  store_whitened<2, 2>(W, J, Ja);
}
__device__ __forceinline__ void lin_row_pose2pose2(const double* mu, const double* W, const double* xa, const double* xb,
                                 double* r, double* Ja, double* Jb) {
  const Se2 P = se2_from_coords(xa[0], xa[1], xa[2]);
  const Se2 Q = se2_from_coords(xb[0], xb[1], xb[2]);
  const double zx = mu[0], zy = mu[1];
  double sz, cz; fast_sincos(mu[2], &sz, &cz);
  double rr[3]; residual_pose2pose2(zx, zy, cz, sz, P, Q, rr);
  const double JA[3][3] = {{1, 0, -P.s * zx - P.c * zy}, {0, 1, P.c * zx - P.s * zy}, {0, 0, 1}};
  const double JB[3][3] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
  store_whitened_vec<3>(W, rr, r);
  store_whitened<3, 3>(W, JA, Ja);
  store_whitened<3, 3>(W, JB, Jb);
}
__device__ __forceinline__ void lin_row_bearingrange(const double* mu, const double* W, const double* xa, const double* xb,
                                   double* r, double* Ja, double* Jb) {
  const Se2 P = se2_from_coords(xa[0], xa[1], xa[2]);
  const double lx = xb[0], ly = xb[1];
  double rr[2]; residual_bearingrange(mu[0], mu[1], P, lx, ly, rr);
  const double dx = lx - P.x, dy = ly - P.y;
  const double plx = P.c * dx + P.s * dy, ply = P.c * dy - P.s * dx;
  const double n2 = plx * plx + ply * ply, n = fast_sqrt(n2);
  const double a00 = ply / n2, a01 = -plx / n2, a10 = -plx / n, a11 = -ply / n;  // A = ∂r/∂pl
  // ∂pl/∂l = Rᵀ = [c s; -s c] ; ∂pl/∂t = -Rᵀ ; ∂pl/∂θ = (pl_y, -pl_x) -> A·that = (1, 0)
  const double l00 = a00 * P.c - a01 * P.s, l01 = a00 * P.s + a01 * P.c;
  const double l10 = a10 * P.c - a11 * P.s, l11 = a10 * P.s + a11 * P.c;
  const double JA[2][3] = {{-l00, -l01, 1.0}, {-l10, -l11, 0.0}};
  const double JB[2][2] = {{l00, l01}, {l10, l11}};
  store_whitened_vec<2>(W, rr, r);
  store_whitened<2, 3>(W, JA, Ja);
  store_whitened<2, 2>(W, JB, Jb);
}
__device__ __forceinline__ void lin_row_pose3pose3(const double* mu, const double* W, const double* xa, const double* xb,
                                 double* r, double* Ja, double* Jb) {
  Se3 P, Q; se3_from_coords(xa, P); se3_from_coords(xb, Q);
  const double* z = mu;
  double Z[9]; so3_exp(z + 3, Z);
  double rr[6]; residual_pose3pose3(z, Z, P, Q, rr);
  double JA[6][6], JB[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) { JA[i][j] = 0.0; JB[i][j] = 0.0; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { JA[i][i] = 1.0; JB[i][i] = -1.0; }
  // ∂r_t/∂δω_p = -R_p [z_t]×   (col-major R)
  const double zx = z[0], zy = z[1], zz = z[2];
  const double S[3][3] = {{0, -zz, zy}, {zz, 0, -zx}, {-zy, zx, 0}};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      JA[i][3 + j] = -(P.R[i] * S[0][j] + P.R[i + 3] * S[1][j] + P.R[i + 6] * S[2][j]);
  double Jr[3][3], Jl[3][3];
  so3_jinv(&rr[3], +1.0, Jr);
  so3_jinv(&rr[3], -1.0, Jl);
  // ∂r_ω/∂δω_p = J_r⁻¹(r_ω) Zᵀ ; ∂r_ω/∂δω_q = -J_l⁻¹(r_ω)
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      JA[3 + i][3 + j] = Jr[i][0] * Z[j] + Jr[i][1] * Z[j + 3] + Jr[i][2] * Z[j + 6];  // (Zᵀ)[k][j] = Z[j + 3k]
      JB[3 + i][3 + j] = -Jl[i][j];
    }
  store_whitened_vec<6>(W, rr, r);
  store_whitened<6, 6>(W, JA, Ja);
  store_whitened<6, 6>(W, JB, Jb);
}
__device__ __forceinline__ void lin_row_priorpose3(const double* mu, const double* W, const double* xa, double* r, double* Ja) {
  Se3 M, P; se3_from_coords(mu, M); se3_from_coords(xa, P);
  double rr[6]; residual_priorpose3(M, P, rr);
  double J[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) J[i][j] = 0.0;
  double Jl[3][3]; so3_jinv(&rr[3], -1.0, Jl);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    J[i][i] = -1.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) J[3 + i][3 + j] = -Jl[i][j];
  }
  store_whitened_vec<6>(W, rr, r);
  store_whitened<6, 6>(W, J, Ja);
}

struct LinDims { int dz, dr, da, db; };
__host__ inline bool lin_dims(int kind, LinDims& d) {
  switch (kind) {
    case ROME_FACTOR_PRIORPOSE2: d = {3, 3, 3, 0}; return true;
    case ROME_FACTOR_POSE2POSE2: d = {3, 3, 3, 3}; return true;
    case ROME_FACTOR_POSE2POINT2BR: d = {2, 2, 3, 2}; return true;
    case ROME_FACTOR_PRIORPOINT2: d = {2, 2, 2, 0}; return true;
    case ROME_FACTOR_POSE3POSE3: d = {6, 6, 6, 6}; return true;
    case ROME_FACTOR_PRIORPOSE3: d = {6, 6, 6, 0}; return true;
    default: return false;
  }
}

// One wavefront per block of 64 factor rows.  The rows are AoS (54 doubles in, 78 out per Pose3Pose3 factor): a thread walking its own
// row touches one 8-byte word per 432-byte stride.  The block therefore moves whole row RANGES -- contiguous in every array -- between
// HBM and LDS with unit-stride 512-byte wave accesses, and the threads work on LDS rows (odd row pitch: conflict-free for the
// row-per-lane access pattern).
template <int D> struct Pitch { static constexpr int v = (D % 2 == 0) ? D + 1 : D; };
template <int D>
__device__ __forceinline__ void rows_in(const double* __restrict__ g, int base, int rows, double* l) {
  const double* src = g + (size_t)base * D;
  for (int q = threadIdx.x; q < rows * D; q += 64) l[(q / D) * Pitch<D>::v + (q % D)] = src[q];
}
template <int D>
__device__ __forceinline__ void rows_out(double* __restrict__ g, int base, int rows, const double* l) {
  double* dst = g + (size_t)base * D;
  for (int q = threadIdx.x; q < rows * D; q += 64) dst[q] = l[(q / D) * Pitch<D>::v + (q % D)];
}
template <int KIND, int DZ, int DR, int DA, int DB>
__global__ void __launch_bounds__(64) k_lin(int F, const double* mu, const double* W, const double* xa, const double* xb,
                                            double* r, double* Ja, double* Jb) {
  constexpr int PZ = Pitch<DZ>::v, PW = Pitch<DR * DR>::v, PA = Pitch<DA>::v, PB = DB ? Pitch<(DB ? DB : 1)>::v : 0;
  constexpr int PR = Pitch<DR>::v, PJA = Pitch<DR * DA>::v, PJB = DB ? Pitch<(DB ? DR * DB : 1)>::v : 0;
  __shared__ double l_mu[64 * PZ], l_W[64 * PW], l_xa[64 * PA], l_xb[64 * (PB ? PB : 1)];
  __shared__ double l_r[64 * PR], l_Ja[64 * PJA], l_Jb[64 * (PJB ? PJB : 1)];
  const int base = blockIdx.x * 64;
  const int rows = F - base < 64 ? F - base : 64;
  rows_in<DZ>(mu, base, rows, l_mu); rows_in<DR * DR>(W, base, rows, l_W); rows_in<DA>(xa, base, rows, l_xa);
  if constexpr (DB > 0) rows_in<DB>(xb, base, rows, l_xb);
  __syncthreads();
  const int t = threadIdx.x;
  if (t < rows) {
    const double* m_ = l_mu + t * PZ; const double* w_ = l_W + t * PW; const double* a_ = l_xa + t * PA;
    [[maybe_unused]] const double* b_ = l_xb + t * (PB ? PB : 1);
    double* r_ = l_r + t * PR; double* ja = l_Ja + t * PJA; [[maybe_unused]] double* jb = l_Jb + t * (PJB ? PJB : 1);
    if constexpr (KIND == ROME_FACTOR_PRIORPOSE2) lin_row_priorpose2(m_, w_, a_, r_, ja);
    else if constexpr (KIND == ROME_FACTOR_PRIORPOINT2) lin_row_priorpoint2(m_, w_, a_, r_, ja);
    else if constexpr (KIND == ROME_FACTOR_POSE2POSE2) lin_row_pose2pose2(m_, w_, a_, b_, r_, ja, jb);
    else if constexpr (KIND == ROME_FACTOR_POSE2POINT2BR) lin_row_bearingrange(m_, w_, a_, b_, r_, ja, jb);
    else if constexpr (KIND == ROME_FACTOR_POSE3POSE3) lin_row_pose3pose3(m_, w_, a_, b_, r_, ja, jb);
    else lin_row_priorpose3(m_, w_, a_, r_, ja);
  }
  __syncthreads();
  rows_out<DR>(r, base, rows, l_r); rows_out<DR * DA>(Ja, base, rows, l_Ja);
  if constexpr (DB > 0) rows_out<DR * DB>(Jb, base, rows, l_Jb);
}

hipError_t launch_linearize(int kind, int F, const double* mu, const double* W, const double* xa, const double* xb,
                            double* r, double* Ja, double* Jb, hipStream_t s) {
  if (F <= 0) return hipSuccess;
  const dim3 g((F + 63) / 64), b(64);
  switch (kind) {
    case ROME_FACTOR_PRIORPOSE2: hipLaunchKernelGGL((k_lin<ROME_FACTOR_PRIORPOSE2, 3, 3, 3, 0>), g, b, 0, s, F, mu, W, xa, xb, r, Ja, Jb); break;
    case ROME_FACTOR_PRIORPOINT2: hipLaunchKernelGGL((k_lin<ROME_FACTOR_PRIORPOINT2, 2, 2, 2, 0>), g, b, 0, s, F, mu, W, xa, xb, r, Ja, Jb); break;
    case ROME_FACTOR_POSE2POSE2: hipLaunchKernelGGL((k_lin<ROME_FACTOR_POSE2POSE2, 3, 3, 3, 3>), g, b, 0, s, F, mu, W, xa, xb, r, Ja, Jb); break;
    case ROME_FACTOR_POSE2POINT2BR: hipLaunchKernelGGL((k_lin<ROME_FACTOR_POSE2POINT2BR, 2, 2, 3, 2>), g, b, 0, s, F, mu, W, xa, xb, r, Ja, Jb); break;
    case ROME_FACTOR_POSE3POSE3: hipLaunchKernelGGL((k_lin<ROME_FACTOR_POSE3POSE3, 6, 6, 6, 6>), g, b, 0, s, F, mu, W, xa, xb, r, Ja, Jb); break;
    case ROME_FACTOR_PRIORPOSE3: hipLaunchKernelGGL((k_lin<ROME_FACTOR_PRIORPOSE3, 6, 6, 6, 0>), g, b, 0, s, F, mu, W, xa, xb, r, Ja, Jb); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace rome

// host-pointer and device-pointer C entry points live in rome_capi.hip (they need the context internals)
