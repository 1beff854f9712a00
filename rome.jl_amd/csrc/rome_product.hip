// rome_product.hip -- belief statistics and the proposal product.
//
//  * k_belief_stats<D>: manifold mean + per-coordinate std of every belief (N particles), the PPE-style summary
//    (`getPPE(...).suggested`, examples/ManhattanBatchAnalysis.jl:59-62) and the input of the KDE bandwidths;
//    same definition as the inflation spread (tangent coordinates about particle 0).
//  * k_product<D>: new belief of a variable from its K proposals -- a regularised importance-sampling product
//    of the K kernel density estimates.  This is a STAND-IN for ApproxManifoldProducts.manifoldProduct
//    (unvendored multiscale Gibbs product): the definition is the one in
//    oracle/rome_oracle.c (ro_product) and is only claimed statistically against the reference.
// One wave (64-thread block) per variable; kernel points of one proposal are staged in LDS and read back with
// wave-uniform (broadcast) addresses; weights are scanned in particle order so the resampling picks are the
// same as a sequential CPU scan.
#include "rome_device_math.hpp"
#include "rome_kernels.h"

namespace rome {

constexpr int kProdMaxN = 512;
constexpr int kProdSlots = kProdMaxN / 64;

template <int D>
__device__ __forceinline__ double tangent_diff(int k, double a, double b) {
  const double d = a - b;
  return (D == 3 && k == 2) ? wrap_pi(d) : d;
}

// per-coordinate (mean offset, std) about particle 0 of one SoA block; all lanes get the result
template <int D>
__device__ __forceinline__ void block_stats(const double* __restrict__ P, int N, double inv_n, double inv_nm1, int lane,
                                            double (&x0)[D], double (&moff)[D], double (&sd)[D]) {
#pragma unroll
  for (int k = 0; k < D; ++k) x0[k] = P[k * N];
  double s[2 * D];
#pragma unroll
  for (int j = 0; j < 2 * D; ++j) s[j] = 0.0;
  for (int i = lane; i < N; i += 64) {
#pragma unroll
    for (int k = 0; k < D; ++k) { const double d = tangent_diff<D>(k, P[k * N + i], x0[k]); s[2 * k] += d; s[2 * k + 1] += d * d; }
  }
  wave_sum_n<2 * D>(s);
#pragma unroll
  for (int k = 0; k < D; ++k) {
    moff[k] = s[2 * k] * inv_n;
    sd[k] = fast_sqrt(fmax(0.0, (s[2 * k + 1] - s[2 * k] * s[2 * k] * inv_n) * inv_nm1));
  }
}

template <int D>
__global__ void __launch_bounds__(64) k_belief_stats(int V, int N, double inv_n, double inv_nm1, const double* __restrict__ bel,
                                                     double* __restrict__ mean, double* __restrict__ sdev) {
  const int v = blockIdx.x;
  if (v >= V) return;
  const int lane = threadIdx.x;
  const double* P = bel + (size_t)v * D * N;
  if constexpr (D == 6) {
    // SE(3): translations as above; rotation: d_i = Log(R_0ᵀ R_i), mean = R_0 Exp(mean d)
    double c0[6], R0[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) c0[k] = P[k * N];
    so3_exp(c0 + 3, R0);
    double s[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) s[j] = 0.0;
    for (int i = lane; i < N; i += 64) {
      double w[3] = {P[3 * N + i], P[4 * N + i], P[5 * N + i]}, R[9], U[9], d[6];
      so3_exp(w, R); mat3_tmul(R0, R, U); so3_log(U, d + 3);
      d[0] = P[i] - c0[0]; d[1] = P[N + i] - c0[1]; d[2] = P[2 * N + i] - c0[2];
#pragma unroll
      for (int k = 0; k < 6; ++k) { s[2 * k] += d[k]; s[2 * k + 1] += d[k] * d[k]; }
    }
    wave_sum_n<12>(s);
    double md[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) md[k] = s[2 * k] * inv_n;
    double E[9], Rm[9], wm[3];
    so3_exp(md + 3, E); mat3_mul(R0, E, Rm); so3_log(Rm, wm);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { mean[6 * v + k] = c0[k] + md[k]; mean[6 * v + 3 + k] = wm[k]; }
#pragma unroll
      for (int k = 0; k < 6; ++k) sdev[6 * v + k] = fast_sqrt(fmax(0.0, (s[2 * k + 1] - s[2 * k] * s[2 * k] * inv_n) * inv_nm1));
    }
  } else {
    double x0[D], moff[D], sd[D];
    block_stats<D>(P, N, inv_n, inv_nm1, lane, x0, moff, sd);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < D; ++k) { mean[D * v + k] = x0[k] + moff[k]; sdev[D * v + k] = sd[k]; }
    }
  }
}

struct ProductArgs {
  int V, N;
  const int32_t* prop_ptr;   // [V+1]
  const int32_t* prop_rows;  // rows of `prop` targeting each variable
  const double* prop;        // [rows][D][N]
  const double* bel_in;      // [V][D][N]
  double* bel_out;           // [V][D][N]
  double inv_n, inv_nm1, c_n;  // c_n: Silverman factor (4/((d+2)N))^(1/(d+4)), host-computed
  uint64_t seed, stream_offset;
};

template <int D>
__global__ void __launch_bounds__(64) k_product(const ProductArgs a) {
  __shared__ double pts[D][kProdMaxN];
  __shared__ double wts[kProdMaxN];
  const int v = blockIdx.x;
  if (v >= a.V) return;
  const int lane = threadIdx.x, N = a.N;
  const int r0 = a.prop_ptr[v], K = a.prop_ptr[v + 1] - r0;
  double* ob = a.bel_out + (size_t)v * D * N;
  if (K <= 1) {
    const double* src = K == 0 ? a.bel_in + (size_t)v * D * N : a.prop + (size_t)a.prop_rows[r0] * D * N;
    for (int i = lane; i < D * N; i += 64) ob[i] = src[i];
    return;
  }
  // pass 1: bandwidths, base proposal, product bandwidth
  int base = 0;
  double best = __builtin_inf();
  double hp_acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) hp_acc[k] = 0.0;
  for (int l = 0; l < K; ++l) {
    const double* P = a.prop + (size_t)a.prop_rows[r0 + l] * D * N;
    double x0[D], moff[D], sd[D];
    block_stats<D>(P, N, a.inv_n, a.inv_nm1, lane, x0, moff, sd);
    double ln = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) { const double h = fmax(a.c_n * sd[k], 1e-6); ln += log(h); hp_acc[k] += 1.0 / (h * h); }
    if (ln < best) { best = ln; base = l; }
  }
  const double* Pb = a.prop + (size_t)a.prop_rows[r0 + base] * D * N;
  // pass 2: log weights of the base particles (lane owns particles lane, lane+64, ...; N <= 512 -> 8 slots)
  constexpr int S = kProdSlots;
  double x[S][D], lw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = lane + 64 * s;
    lw[s] = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) x[s][k] = Pb[k * N + (i < N ? i : 0)];
  }
  for (int l = 0; l < K; ++l) {
    if (l == base) continue;
    const double* P = a.prop + (size_t)a.prop_rows[r0 + l] * D * N;
    double x0[D], moff[D], sd[D], ih[D];
    block_stats<D>(P, N, a.inv_n, a.inv_nm1, lane, x0, moff, sd);
#pragma unroll
    for (int k = 0; k < D; ++k) ih[k] = 1.0 / fmax(a.c_n * sd[k], 1e-6);
    __syncthreads();
    for (int i = lane; i < N; i += 64) {
#pragma unroll
      for (int k = 0; k < D; ++k) pts[k][i] = P[k * N + i];
    }
    __syncthreads();
    double qmin[S], sacc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { qmin[s] = __builtin_inf(); sacc[s] = 0.0; }
    for (int j = 0; j < N; ++j) {
      double y[D];
#pragma unroll
      for (int k = 0; k < D; ++k) y[k] = pts[k][j];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) { const double d = tangent_diff<D>(k, x[s][k], y[k]) * ih[k]; q += d * d; }
        qmin[s] = fmin(qmin[s], q);
      }
    }
    for (int j = 0; j < N; ++j) {
      double y[D];
#pragma unroll
      for (int k = 0; k < D; ++k) y[k] = pts[k][j];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if (lane + 64 * s < N) {  // skip idle slots (keeps exp() off them)
          double q = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) { const double d = tangent_diff<D>(k, x[s][k], y[k]) * ih[k]; q += d * d; }
          sacc[s] += exp(-0.5 * (q - qmin[s]));
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) if (lane + 64 * s < N) lw[s] += -0.5 * qmin[s] + log(sacc[s]);
  }
  // normalise, publish weights in particle order
  double mx = -__builtin_inf();
#pragma unroll
  for (int s = 0; s < S; ++s) if (lane + 64 * s < N) mx = fmax(mx, lw[s]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  __syncthreads();
#pragma unroll
  for (int s = 0; s < S; ++s) if (lane + 64 * s < N) wts[lane + 64 * s] = exp(lw[s] - mx);
  __syncthreads();
  // systematic resampling: sequential scan in particle order (same arithmetic order as a CPU loop)
  double T = 0.0;
  for (int m = 0; m < N; ++m) T += wts[m];
  const uint64_t stream = a.stream_offset + (uint64_t)v;
  const u32x4 uw = philox4x32_10(u32x4{0xFFFFFFFFu, (uint32_t)stream, (uint32_t)(stream >> 32), (3u << 16)},
                                 (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const double u = ((double)uw.x + 0.5) * (1.0 / 4294967296.0);
  double tau[S]; int pick[S]; bool found[S];
#pragma unroll
  for (int s = 0; s < S; ++s) { tau[s] = ((double)(lane + 64 * s) + u) * T / (double)N; pick[s] = N - 1; found[s] = false; }
  double cum = 0.0;
  for (int m = 0; m < N; ++m) {
    cum += wts[m];
#pragma unroll
    for (int s = 0; s < S; ++s) if (!found[s] && cum > tau[s]) { pick[s] = m; found[s] = true; }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = lane + 64 * s;
    if (i < N) {
      double xi[D];
      rng_normals<D>(a.seed, stream, (uint32_t)i, xi);
      double o[D];
#pragma unroll
      for (int k = 0; k < D; ++k) o[k] = Pb[k * N + pick[s]] + xi[k] / fast_sqrt(hp_acc[k]);
      if constexpr (D == 3) o[2] = wrap_pi(o[2]);
#pragma unroll
      for (int k = 0; k < D; ++k) ob[k * N + i] = o[k];
    }
  }
}

hipError_t launch_belief_stats(int dim, int V, int N, const double* bel, double* mean, double* sdev, hipStream_t s) {
  if (V <= 0) return hipSuccess;
  const double inv_n = 1.0 / N, inv_nm1 = N > 1 ? 1.0 / (N - 1) : 1.0;
  switch (dim) {
    case 2: hipLaunchKernelGGL(k_belief_stats<2>, dim3(V), dim3(64), 0, s, V, N, inv_n, inv_nm1, bel, mean, sdev); break;
    case 3: hipLaunchKernelGGL(k_belief_stats<3>, dim3(V), dim3(64), 0, s, V, N, inv_n, inv_nm1, bel, mean, sdev); break;
    case 6: hipLaunchKernelGGL(k_belief_stats<6>, dim3(V), dim3(64), 0, s, V, N, inv_n, inv_nm1, bel, mean, sdev); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_product(int dim, int V, int N, const int32_t* prop_ptr, const int32_t* prop_rows, const double* prop,
                          const double* bel_in, double* bel_out, double c_n, uint64_t seed, uint64_t stream_offset, hipStream_t s) {
  if (V <= 0) return hipSuccess;
  if (N > kProdMaxN) return hipErrorInvalidValue;
  ProductArgs a;
  a.V = V; a.N = N; a.prop_ptr = prop_ptr; a.prop_rows = prop_rows; a.prop = prop; a.bel_in = bel_in; a.bel_out = bel_out;
  a.inv_n = 1.0 / N; a.inv_nm1 = N > 1 ? 1.0 / (N - 1) : 1.0; a.c_n = c_n; a.seed = seed; a.stream_offset = stream_offset;
  switch (dim) {
    case 2: hipLaunchKernelGGL(k_product<2>, dim3(V), dim3(64), 0, s, a); break;
    case 3: hipLaunchKernelGGL(k_product<3>, dim3(V), dim3(64), 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace rome
