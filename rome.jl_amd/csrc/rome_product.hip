// rome_product.hip -- belief statistics and the proposal product.
//
//  * k_belief_stats<D>: manifold mean + per-coordinate std of every belief (N particles), the PPE-style summary
//    (`getPPE(...).suggested`, examples/ManhattanBatchAnalysis.jl:59-62) and the input of the KDE bandwidths;
//    same definition as the inflation spread (tangent coordinates about particle 0).
//  * k_product<D>: new belief of a variable from its K proposals -- a regularised importance-sampling product
//    of the K kernel density estimates.  This is a STAND-IN for ApproxManifoldProducts.manifoldProduct
//    (unvendored multiscale Gibbs product): the definition is the one in
//    oracle/rome_oracle.c (ro_product) and is only claimed statistically against the reference.
// Belief statistics: one wave (64-thread block) per variable.  Product: one 256-thread block per variable (see
// k_product); weights are scanned in particle order so the resampling picks are the same as a sequential CPU scan.
#include "rome_device_math.hpp"
#include "rome_kernels.h"

namespace rome {

constexpr int kProdMaxN = 512;

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
  const double* prop_bw;     // [rows][D] kernel bandwidths of the proposals (rome_kde_bandwidth_dev), or null: Silverman in-kernel
  const double* bel_in;      // [V][D][N]
  double* bel_out;           // [V][D][N]
  double inv_n, inv_nm1, c_n;  // c_n: Silverman factor (4/((d+2)N))^(1/(d+4)), host-computed
  uint64_t seed, stream_offset;
};

// One 256-thread block (4 wavefronts) per variable.  The K-1 non-base proposals are dealt round-robin to the four
// waves -- a variable with many proposals (loop-closure hubs: K up to ~11 on Manhattan) no longer serialises them in
// one wave.  Inside a wave, lane l owns base particles l, l+64, ... (S slots, N <= 64·S) and walks the N kernel points
// of its proposal with wave-uniform (scalar-cache) loads; the per-proposal log-weight contributions go to LDS and are
// summed per particle in proposal order, so the arithmetic order -- and the resampling picks -- are those of the
// sequential definition in oracle/rome_oracle.c (ro_product).
constexpr int kProdWaves = 4;
constexpr int kProdChunk = 8;   // proposals whose contributions are buffered in LDS at a time
constexpr int kProdMaxK = 32;   // proposals whose bandwidths are kept in LDS (more: recomputed where needed)

// kernel bandwidths of one proposal: caller-supplied (leave-one-out likelihood rule, rome_kde.hip) or Silverman's rule on
// the proposal's spread; floored at 1e-6 either way
template <int D>
__device__ __forceinline__ void proposal_bandwidth(const ProductArgs& a, int row, const double* __restrict__ P, int N, int lane,
                                                   double (&h)[D]) {
  if (a.prop_bw) {
#pragma unroll
    for (int k = 0; k < D; ++k) h[k] = fmax(a.prop_bw[(size_t)row * D + k], 1e-6);   // wave-uniform address
  } else {
    double x0[D], moff[D], sd[D];
    block_stats<D>(P, N, a.inv_n, a.inv_nm1, lane, x0, moff, sd);
#pragma unroll
    for (int k = 0; k < D; ++k) h[k] = fmax(a.c_n * sd[k], 1e-6);
  }
}

template <int D, int S>
__global__ void __launch_bounds__(64 * kProdWaves) k_product(const ProductArgs a) {
  constexpr int T4 = (S + kProdWaves - 1) / kProdWaves;   // particles per thread in the block-wide phases
  constexpr bool kStage = S <= 4;   // N <= 256: every wave stages the points of its proposal in its own LDS region
  __shared__ double pts[kStage ? kProdWaves : 1][D][kStage ? 64 * S : 1];
  __shared__ double contrib[kProdChunk][64 * S];
  __shared__ double wts[64 * S];
  __shared__ double red[kProdWaves];
  __shared__ double ihbuf[kProdMaxK][D];
  __shared__ double lnbuf[kProdMaxK];
  const int v = blockIdx.x;
  if (v >= a.V) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), N = a.N;
  const int r0 = a.prop_ptr[v], K = a.prop_ptr[v + 1] - r0;
  double* ob = a.bel_out + (size_t)v * D * N;
  if (K <= 1) {
    const double* src = K == 0 ? a.bel_in + (size_t)v * D * N : a.prop + (size_t)a.prop_rows[r0] * D * N;
    for (int i = tid; i < D * N; i += 64 * kProdWaves) ob[i] = src[i];
    return;
  }
  // pass 1: bandwidths (proposals dealt to the waves, exchanged through LDS), base proposal, product bandwidth
  int base = 0;
  double best = __builtin_inf();
  double hp_acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) hp_acc[k] = 0.0;
  for (int l0 = 0; l0 < K; l0 += kProdMaxK) {
    const int cnt = min(kProdMaxK, K - l0);
    for (int c = wave; c < cnt; c += kProdWaves) {
      const int row = a.prop_rows[r0 + l0 + c];
      double h[D];
      proposal_bandwidth<D>(a, row, a.prop + (size_t)row * D * N, N, lane, h);
      double ln = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        ln += fast_log(h[k]);
        if (lane == 0) ihbuf[c][k] = 1.0 / h[k];
      }
      if (lane == 0) lnbuf[c] = ln;
    }
    __syncthreads();
    for (int c = 0; c < cnt; ++c) {
#pragma unroll
      for (int k = 0; k < D; ++k) { const double ih = ihbuf[c][k]; hp_acc[k] = fma(ih, ih, hp_acc[k]); }
      const double ln = lnbuf[c];
      if (ln < best) { best = ln; base = l0 + c; }
    }
    if (l0 + kProdMaxK < K) __syncthreads();   // the buffers are reused by the next chunk
  }
  const bool h_cached = K <= kProdMaxK;        // single chunk: ihbuf still holds every proposal's 1/h
  const double* __restrict__ Pb = a.prop + (size_t)a.prop_rows[r0 + base] * D * N;
  // pass 2: log weights of the base particles
  double x[S][D];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = lane + 64 * s;
#pragma unroll
    for (int k = 0; k < D; ++k) x[s][k] = Pb[k * N + (i < N ? i : 0)];
  }
  double lw[T4];
#pragma unroll
  for (int s = 0; s < T4; ++s) lw[s] = 0.0;
  for (int c0 = 0; c0 < K - 1; c0 += kProdChunk) {          // chunk of non-base proposals c0 .. c0+cnt-1
    const int cnt = min(kProdChunk, K - 1 - c0);
    for (int c = wave; c < cnt; c += kProdWaves) {
      const int nb = c0 + c, l = nb < base ? nb : nb + 1;   // nb-th non-base proposal
      const int row = a.prop_rows[r0 + l];
      const double* __restrict__ P = a.prop + (size_t)row * D * N;
      double ih[D];
      if (h_cached) {
#pragma unroll
        for (int k = 0; k < D; ++k) ih[k] = ihbuf[l][k];
      } else {
        double h[D];
        proposal_bandwidth<D>(a, row, P, N, lane, h);
#pragma unroll
        for (int k = 0; k < D; ++k) ih[k] = 1.0 / h[k];
      }
      double qmin[S], sacc[S];
#pragma unroll
      for (int s = 0; s < S; ++s) { qmin[s] = __builtin_inf(); sacc[s] = 0.0; }
      // one pass, running log-sum-exp: sacc = Σ_j exp(-½(q_j - qmin)) with qmin the smallest q seen so far
      if constexpr (kStage) {   // wave-private LDS region: only wave-level ordering is needed
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int i = lane + 64 * s;
          if (i < N) {
#pragma unroll
            for (int k = 0; k < D; ++k) pts[wave][k][i] = P[k * N + i];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      }
      for (int j = 0; j < N; ++j) {
        double y[D];
#pragma unroll
        for (int k = 0; k < D; ++k) y[k] = kStage ? pts[kStage ? wave : 0][k][kStage ? j : 0] : P[k * N + j];   // wave-uniform address
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if (lane + 64 * s < N) {  // skip idle slots (keeps exp() off them)
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) { const double d = tangent_diff<D>(k, x[s][k], y[k]) * ih[k]; q += d * d; }
            const double dq = q - qmin[s];
            const double e = fast_exp_neg(-0.5 * fabs(dq));
            sacc[s] = dq < 0.0 ? fma(sacc[s], e, 1.0) : sacc[s] + e;
            qmin[s] = fmin(qmin[s], q);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) if (lane + 64 * s < N) contrib[c][lane + 64 * s] = -0.5 * qmin[s] + fast_log(sacc[s]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < T4; ++s) {
      const int i = tid + 64 * kProdWaves * s;
      if (i < N) for (int c = 0; c < cnt; ++c) lw[s] += contrib[c][i];   // proposal order
    }
    __syncthreads();
  }
  // normalise, publish weights in particle order
  double mx = -__builtin_inf();
#pragma unroll
  for (int s = 0; s < T4; ++s) if (tid + 64 * kProdWaves * s < N) mx = fmax(mx, lw[s]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
#pragma unroll
  for (int s = 0; s < T4; ++s) { const int i = tid + 64 * kProdWaves * s; if (i < N) wts[i] = exp(lw[s] - mx); }
  __syncthreads();
  // systematic resampling: cumulative weights by a sequential scan in particle order (one wave; same arithmetic
  // order as a CPU loop), then every thread looks its picks up by bisection (first m with cum[m] > τ, else N-1)
  if (wave == 0) {
    double cum = 0.0;
    for (int m = 0; m < N; ++m) {
      cum += wts[m];
      if (lane == (m & 63)) contrib[0][m] = cum;   // contrib is free again: reuse its first row
    }
  }
  __syncthreads();
  const double* cumw = contrib[0];
  const double T = cumw[N - 1];
  const uint64_t stream = a.stream_offset + (uint64_t)v;
  const u32x4 uw = philox4x32_10(u32x4{0xFFFFFFFFu, (uint32_t)stream, (uint32_t)(stream >> 32), (3u << 16)},
                                 (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const double u = ((double)uw.x + 0.5) * (1.0 / 4294967296.0);
  int pick[T4];
#pragma unroll
  for (int s = 0; s < T4; ++s) {
    const double tau = ((double)(tid + 64 * kProdWaves * s) + u) * T / (double)N;
    int lo = 0, hi = N - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cumw[mid] > tau) hi = mid; else lo = mid + 1;
    }
    pick[s] = lo;
  }
#pragma unroll
  for (int s = 0; s < T4; ++s) {
    const int i = tid + 64 * kProdWaves * s;
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

// ---- SE(3) product (Pose3 beliefs; coordinates [t(3); rotation vector(3)]) ------------------------------------------------
// Same definition as k_product (oracle: ro_product_bw, dim 6) with the tangent difference of SE(3):
//   d(x, y) = (x.t − y.t, Log(R_yᵀ R_x)),   jitter  t += h_t ⊙ ξ_t,  R ← R Exp(h_ω ⊙ ξ_ω).
// Rotations are unit quaternions from load to store (one Exp per point when it is loaded or staged, one Log per pair).
struct Se3Pt { double t[3], q[4]; };
__device__ __forceinline__ void se3_load(const double* __restrict__ P, int N, int i, Se3Pt& a) {
  double w[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { a.t[k] = P[k * N + i]; w[k] = P[(3 + k) * N + i]; }
  quat_exp(w, a.q);
}
__device__ __forceinline__ void se3_diff(const Se3Pt& x, const Se3Pt& y, double (&d)[6]) {
  double e[4];
  quat_cmul(y.q, x.q, e);
  quat_log(e, d + 3);
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = x.t[k] - y.t[k];
}
__device__ __forceinline__ void proposal_bandwidth_se3(const ProductArgs& a, int row, const double* __restrict__ P, int N, int lane,
                                                       double (&h)[6]) {
  if (a.prop_bw) {
#pragma unroll
    for (int k = 0; k < 6; ++k) h[k] = fmax(a.prop_bw[(size_t)row * 6 + k], 1e-6);
    return;
  }
  Se3Pt x0; se3_load(P, N, 0, x0);
  double s[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) s[j] = 0.0;
  for (int i = lane; i < N; i += 64) {
    Se3Pt xi; se3_load(P, N, i, xi);
    double d[6]; se3_diff(xi, x0, d);
#pragma unroll
    for (int k = 0; k < 6; ++k) { s[2 * k] += d[k]; s[2 * k + 1] += d[k] * d[k]; }
  }
  wave_sum_n<12>(s);
#pragma unroll
  for (int k = 0; k < 6; ++k)
    h[k] = fmax(a.c_n * fast_sqrt(fmax(0.0, (s[2 * k + 1] - s[2 * k] * s[2 * k] * a.inv_n) * a.inv_nm1)), 1e-6);
}

#ifndef ROME_PROD3_MINBLK
#define ROME_PROD3_MINBLK 3   // blocks per CU: 3 waves/SIMD (168 VGPRs, a few spills) measured 1.67 ms vs 2.10 (2) and 2.49 (4) on the 10^4-pose helix
#endif
template <int S>
__global__ void __launch_bounds__(64 * kProdWaves, ROME_PROD3_MINBLK) k_product_se3(const ProductArgs a) {
  constexpr int D = 6;
  constexpr int T4 = (S + kProdWaves - 1) / kProdWaves;
  __shared__ double pts[kProdWaves][7][64 * S];   // every wave stages (t, q) of the points of its proposal
  __shared__ double contrib[kProdChunk][64 * S];
  __shared__ double wts[64 * S];
  __shared__ double red[kProdWaves];
  __shared__ double ihbuf[kProdMaxK][D];
  __shared__ double lnbuf[kProdMaxK];
  const int v = blockIdx.x;
  if (v >= a.V) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), N = a.N;
  const int r0 = a.prop_ptr[v], K = a.prop_ptr[v + 1] - r0;
  double* ob = a.bel_out + (size_t)v * D * N;
  if (K <= 1) {
    const double* src = K == 0 ? a.bel_in + (size_t)v * D * N : a.prop + (size_t)a.prop_rows[r0] * D * N;
    for (int i = tid; i < D * N; i += 64 * kProdWaves) ob[i] = src[i];
    return;
  }
  int base = 0;
  double best = __builtin_inf();
  double hp_acc[D];
#pragma unroll
  for (int k = 0; k < D; ++k) hp_acc[k] = 0.0;
  for (int l0 = 0; l0 < K; l0 += kProdMaxK) {
    const int cnt = min(kProdMaxK, K - l0);
    for (int c = wave; c < cnt; c += kProdWaves) {
      const int row = a.prop_rows[r0 + l0 + c];
      double h[D];
      proposal_bandwidth_se3(a, row, a.prop + (size_t)row * D * N, N, lane, h);
      double ln = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        ln += fast_log(h[k]);
        if (lane == 0) ihbuf[c][k] = 1.0 / h[k];
      }
      if (lane == 0) lnbuf[c] = ln;
    }
    __syncthreads();
    for (int c = 0; c < cnt; ++c) {
#pragma unroll
      for (int k = 0; k < D; ++k) { const double ih = ihbuf[c][k]; hp_acc[k] = fma(ih, ih, hp_acc[k]); }
      const double ln = lnbuf[c];
      if (ln < best) { best = ln; base = l0 + c; }
    }
    if (l0 + kProdMaxK < K) __syncthreads();
  }
  const bool h_cached = K <= kProdMaxK;
  const double* __restrict__ Pb = a.prop + (size_t)a.prop_rows[r0 + base] * D * N;
  Se3Pt x[S];
#pragma unroll
  for (int s = 0; s < S; ++s) { const int i = lane + 64 * s; se3_load(Pb, N, i < N ? i : 0, x[s]); }
  double lw[T4];
#pragma unroll
  for (int s = 0; s < T4; ++s) lw[s] = 0.0;
  for (int c0 = 0; c0 < K - 1; c0 += kProdChunk) {
    const int cnt = min(kProdChunk, K - 1 - c0);
    for (int c = wave; c < cnt; c += kProdWaves) {
      const int nb = c0 + c, l = nb < base ? nb : nb + 1;
      const int row = a.prop_rows[r0 + l];
      const double* __restrict__ P = a.prop + (size_t)row * D * N;
      double ih[D];
      if (h_cached) {
#pragma unroll
        for (int k = 0; k < D; ++k) ih[k] = ihbuf[l][k];
      } else {
        double h[D];
        proposal_bandwidth_se3(a, row, P, N, lane, h);
#pragma unroll
        for (int k = 0; k < D; ++k) ih[k] = 1.0 / h[k];
      }
      double qmin[S], sacc[S];
#pragma unroll
      for (int s = 0; s < S; ++s) { qmin[s] = __builtin_inf(); sacc[s] = 0.0; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int i = lane + 64 * s;
        if (i < N) {
          Se3Pt y; se3_load(P, N, i, y);
#pragma unroll
          for (int k = 0; k < 3; ++k) pts[wave][k][i] = y.t[k];
#pragma unroll
          for (int k = 0; k < 4; ++k) pts[wave][3 + k][i] = y.q[k];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      for (int j = 0; j < N; ++j) {
        Se3Pt y;
#pragma unroll
        for (int k = 0; k < 3; ++k) y.t[k] = pts[wave][k][j];       // wave-uniform address
#pragma unroll
        for (int k = 0; k < 4; ++k) y.q[k] = pts[wave][3 + k][j];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if (lane + 64 * s < N) {
            double d[6];
            se3_diff(x[s], y, d);
            double q = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) { const double dt = d[k] * ih[k]; q += dt * dt; const double dw = d[3 + k] * ih[3 + k]; q += dw * dw; }
            const double dq = q - qmin[s];
            const double e = fast_exp_neg(-0.5 * fabs(dq));
            sacc[s] = dq < 0.0 ? fma(sacc[s], e, 1.0) : sacc[s] + e;
            qmin[s] = fmin(qmin[s], q);
          }
        }
      }
#pragma unroll
      for (int s = 0; s < S; ++s) if (lane + 64 * s < N) contrib[c][lane + 64 * s] = -0.5 * qmin[s] + fast_log(sacc[s]);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < T4; ++s) {
      const int i = tid + 64 * kProdWaves * s;
      if (i < N) for (int c = 0; c < cnt; ++c) lw[s] += contrib[c][i];
    }
    __syncthreads();
  }
  double mx = -__builtin_inf();
#pragma unroll
  for (int s = 0; s < T4; ++s) if (tid + 64 * kProdWaves * s < N) mx = fmax(mx, lw[s]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
#pragma unroll
  for (int s = 0; s < T4; ++s) { const int i = tid + 64 * kProdWaves * s; if (i < N) wts[i] = exp(lw[s] - mx); }
  __syncthreads();
  if (wave == 0) {
    double cum = 0.0;
    for (int m = 0; m < N; ++m) {
      cum += wts[m];
      if (lane == (m & 63)) contrib[0][m] = cum;
    }
  }
  __syncthreads();
  const double* cumw = contrib[0];
  const double T = cumw[N - 1];
  const uint64_t stream = a.stream_offset + (uint64_t)v;
  const u32x4 uw = philox4x32_10(u32x4{0xFFFFFFFFu, (uint32_t)stream, (uint32_t)(stream >> 32), (3u << 16)},
                                 (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const double u = ((double)uw.x + 0.5) * (1.0 / 4294967296.0);
#pragma unroll
  for (int s = 0; s < T4; ++s) {
    const int i = tid + 64 * kProdWaves * s;
    if (i < N) {
      const double tau = ((double)i + u) * T / (double)N;
      int lo = 0, hi = N - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cumw[mid] > tau) hi = mid; else lo = mid + 1;
      }
      Se3Pt p; se3_load(Pb, N, lo, p);
      double xi[D];
      rng_normals<D>(a.seed, stream, (uint32_t)i, xi);
      double e[3], qe[4], qn[4], wn[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) e[k] = xi[3 + k] / fast_sqrt(hp_acc[3 + k]);
      quat_exp(e, qe); quat_mul(p.q, qe, qn); quat_log(qn, wn);
#pragma unroll
      for (int k = 0; k < 3; ++k) { ob[k * N + i] = p.t[k] + xi[k] / fast_sqrt(hp_acc[k]); ob[(3 + k) * N + i] = wn[k]; }
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
                          const double* prop_bw, const double* bel_in, double* bel_out, double c_n, uint64_t seed,
                          uint64_t stream_offset, hipStream_t s) {
  if (V <= 0) return hipSuccess;
  if (N > kProdMaxN) return hipErrorInvalidValue;
  ProductArgs a;
  a.V = V; a.N = N; a.prop_ptr = prop_ptr; a.prop_rows = prop_rows; a.prop = prop; a.prop_bw = prop_bw; a.bel_in = bel_in; a.bel_out = bel_out;
  a.inv_n = 1.0 / N; a.inv_nm1 = N > 1 ? 1.0 / (N - 1) : 1.0; a.c_n = c_n; a.seed = seed; a.stream_offset = stream_offset;
  if (dim == 6) {   // Pose3: quaternion state, points staged per wave -- N <= 256
    if (N <= 64) hipLaunchKernelGGL((k_product_se3<1>), dim3(V), dim3(64 * kProdWaves), 0, s, a);
    else if (N <= 128) hipLaunchKernelGGL((k_product_se3<2>), dim3(V), dim3(64 * kProdWaves), 0, s, a);
    else if (N <= 256) hipLaunchKernelGGL((k_product_se3<4>), dim3(V), dim3(64 * kProdWaves), 0, s, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (dim != 2 && dim != 3) return hipErrorInvalidValue;
#define ROME_LAUNCH_PRODUCT(S) \
  do { if (dim == 2) hipLaunchKernelGGL((k_product<2, S>), dim3(V), dim3(64 * kProdWaves), 0, s, a); \
       else hipLaunchKernelGGL((k_product<3, S>), dim3(V), dim3(64 * kProdWaves), 0, s, a); } while (0)
  if (N <= 64) ROME_LAUNCH_PRODUCT(1);
  else if (N <= 128) ROME_LAUNCH_PRODUCT(2);
  else if (N <= 256) ROME_LAUNCH_PRODUCT(4);
  else ROME_LAUNCH_PRODUCT(8);
#undef ROME_LAUNCH_PRODUCT
  return hipGetLastError();
}

}  // namespace rome
