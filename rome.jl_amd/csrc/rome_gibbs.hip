// rome_gibbs.hip -- manifoldProduct: multiscale Gibbs sampling from a product of kernel density estimates.
//
// What the reference does with the proposals of a variable (⚠AMP `manifoldProduct(ff, manifold; Niter=1)` -> ⚠KDE.jl
// `prodAppxMSGibbsS`, a port of A. Ihler's kde toolbox): Ihler, Sudderth, Freeman, Willsky, "Efficient multiscale sampling from
// products of Gaussian mixtures", NIPS 2003.  Restated from the paper / published implementation (neither package is vendored
// with RoME); the definition, step by step, is ro_product_msgibbs in oracle/rome_oracle.c and
// the two are compared sample by sample (tests/test_gpu_gibbs.py).  Random and unseeded upstream: pinned statistically only.
//
// Mapping on gfx950: three kernels.
//   build   k_gibbs_trees: one wavefront per proposal (four per block) writes the proposal's ball tree (6.9 kB for Pose2) to a
//           workspace in HBM -- every proposal feeds exactly one product.  A tree is built top-down by sorting:
//           at level l every node re-sorts its index range along its widest coordinate -- ONE 128-key bitonic sort of the whole
//           wavefront per level with the key (node, coordinate, id): sorting by node first keeps every point inside its node's
//           range, so the nodes of a level are sorted simultaneously (segment extents by LDS integer min/max atomics).
//           Node statistics bottom-up by Chan's pairwise update with the children fetched by lane shuffles (same arithmetic
//           order as the oracle), stored in single precision as offsets from the proposal's point 0, [coordinate][node].
//   order   k_gibbs_order: the variables by descending number of proposals (one block, counting sort) -- the dispatch order of
//   sample  k_product_gibbs: one block per variable, lane = output sample (128 threads for N <= 128, 256 up to N = 256).  The current level of every tree
//           of the variable is staged in LDS (2.8 kB per tree; the whole trees would pin 81 kB for a hub variable with 11
//           proposals); every categorical draw walks the candidates of a level two at a time with WAVE-UNIFORM node statistics
//           (8-byte LDS broadcasts) against the lane's own point / product Gaussian: no divergence; one-pass reservoir selection
//           (running max of log p, rescaled total) driven by a xorshift32 stream seeded from one Philox word.  The candidate
//           arithmetic is an exact single-precision specification shared with the oracle (below).
#include "rome_device_math.hpp"
#include "rome_kernels.h"

namespace rome {

// Every kernel below is instantiated for NM = 128 (N <= 128: lane = output sample of a 128-thread block, two particles per lane of the
// tree-building wavefront -- the sizes everything was tuned at) and NM = 256 (128 < N <= 256: twice the threads / slots / LDS images).
constexpr int kGibbsMaxN = 256;

struct GibbsArgs {
  int V, N, L, max_k, k_lds, iters;   // k_lds: level images resident in LDS per block (variables with more proposals stream them)
  uint32_t circ;
  const int32_t* prop_ptr; const int32_t* prop_rows;
  const double* prop; const double* prop_bw; const double* bel_in; double* bel_out;
  void* trees;            // workspace: one GibbsTree<D> per proposal row
  const int32_t* order;   // workspace: variables by descending number of proposals
  int n_rows;
  uint64_t seed, stream_offset;
  GibbsPlace place;       // optional indirections (all nullptr: variable v lives in block v, draws stream v, no mirror)
};

template <int D, int NM>
struct alignas(8) GibbsTree {
  double ref[D], h[D];
  double q0[4];   // D = 6 (Pose3): unit quaternion of point 0 -- coordinates 3..5 are Log(q0* ⊗ q_i), the chart at that rotation
  float mean[D][NM], var[D][NM], ivar[D][NM], cz[NM];   // node (l, z) at index 2^l - 1 + z: a level is contiguous per coordinate
  float ys[D][NM];                                      // level L: the single points in sorted order
  float lvar[D], livar[D], lcz;
  int row;
  uint8_t perm[NM];
};

__device__ __forceinline__ double gwrap(double d) { return d - 6.283185307179586476925287 * rint(d * 0.15915494309189533576888); }
__device__ __forceinline__ uint32_t fkey(float f) {   // order-preserving map float -> uint32
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ void node_range(int N, int l, int z, int* a, int* b) {
  *a = (int)(((long long)z * N) >> l); *b = (int)(((long long)(z + 1) * N) >> l);
}
__device__ __forceinline__ double shfl_f64(double v, int src) {
  const int lo = __shfl(__double2loint(v), src, 64), hi = __shfl(__double2hiint(v), src, 64);
  return __hiloint2double(hi, lo);
}

// one wavefront builds tree `T` of proposal `P` ([D][N] doubles, bandwidths h); per-wave LDS scratch ybuf / ext / kb.
// S = NM / 64 positions per lane (2 for N <= 128, 4 up to 256); the NM = 128 instantiation is the code of rounds 2-3, statement by
// statement.
template <int D, int NM>
__device__ void gibbs_build(GibbsTree<D, NM>* __restrict__ T, const double* __restrict__ P, const double* __restrict__ hb, int row,
                            int N, int L, uint32_t circ, int lane, float* __restrict__ ybuf /*[NM][D]*/, int* __restrict__ ext /*[NM/2][2D]*/,
                            uint64_t* __restrict__ kb /*[NM]*/) {
  constexpr int S = NM / 64;
  // ---- offsets from point 0, single precision; positions p = lane, lane + 64, ...
  float y[S][D];
  int id[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = lane + 64 * s;
    id[s] = i;
    const int ii = i < N ? i : 0;
#pragma unroll
    for (int d = 0; d < (D == 6 ? 3 : D); ++d) {
      double o = P[d * N + ii] - P[d * N];
      if ((circ >> d) & 1u) o = gwrap(o);
      y[s][d] = (float)o + 0.0f;   // (-0 -> +0: one order for equal offsets)
    }
    if constexpr (D == 6) {   // Pose3: rotation coordinates in the chart at the rotation of point 0
      const double w0[3] = {P[3 * N], P[4 * N], P[5 * N]}, wi[3] = {P[3 * N + ii], P[4 * N + ii], P[5 * N + ii]};
      double qa[4], qb[4], e[4], lg[3];
      quat_exp(w0, qa); quat_exp(wi, qb); quat_cmul(qa, qb, e); quat_log(e, lg);
#pragma unroll
      for (int k = 0; k < 3; ++k) y[s][3 + k] = (float)lg[k] + 0.0f;
      if (lane == 0 && s == 0) { T->q0[0] = qa[0]; T->q0[1] = qa[1]; T->q0[2] = qa[2]; T->q0[3] = qa[3]; }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) if (i < N) ybuf[i * D + d] = y[s][d];
  }
  if (lane < D) { T->ref[lane] = P[lane * N]; T->h[lane] = fmax(hb[lane], 1e-6); }
  if (lane == 0) T->row = row;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  // ---- top-down: per level the points of every node are ordered by (key along the node's widest coordinate, id)
  for (int l = 0; l < L; ++l) {
    const int nn = 1 << l;
    for (int q = lane; q < nn * 2 * D; q += 64) ext[q] = (q & 1) ? (int)0x80000000 : 0x7FFFFFFF;   // (min, max) per node and coordinate
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    int z[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int p = lane + 64 * s;
      z[s] = p < N ? (int)((((long long)(p + 1) << l) - 1) / N) : nn;   // node of position p (padding positions sort last)
      if (p < N) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const int k = (int)(fkey(y[s][d]) ^ 0x80000000u);    // signed order
          atomicMin(&ext[(z[s] * D + d) * 2], k); atomicMax(&ext[(z[s] * D + d) * 2 + 1], k);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    uint64_t key[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int p = lane + 64 * s;
      if (p < N) {
        int best = 0; float be = -1.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const uint32_t kmn = (uint32_t)ext[(z[s] * D + d) * 2] ^ 0x80000000u, kmx = (uint32_t)ext[(z[s] * D + d) * 2 + 1] ^ 0x80000000u;
          const float mn = __uint_as_float((kmn & 0x80000000u) ? (kmn & 0x7FFFFFFFu) : ~kmn), mx = __uint_as_float((kmx & 0x80000000u) ? (kmx & 0x7FFFFFFFu) : ~kmx);
          const float e = mx - mn;
          if (e > be) { be = e; best = d; }
        }
        float kv = y[s][0];
#pragma unroll
        for (int d = 1; d < D; ++d) kv = best == d ? y[s][d] : kv;
        key[s] = ((uint64_t)z[s] << 40) | ((uint64_t)fkey(kv) << 8) | (uint64_t)id[s];
      } else key[s] = ~0ull;
    }
    // The positions of a node are already its own (the level above ordered by node): ordering WITHIN the node is all that is left, and
    // a point's place is the node's first position + the number of the node's points with a smaller (key, id) -- counted against
    // the node's keys in LDS (lanes of one node read the same address: broadcasts).  Σ_l N / 2^l = 2N compare steps per point
    // instead of the 28 exchange stages of a full 128-key bitonic sort at every level (which this replaced: 223 -> 207 µs per
    // Manhattan sweep of proposals); the same order (keys are unique by id).
    {
      int na[S], nbnd[S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int p = lane + 64 * s;
        na[s] = 0; nbnd[s] = 0;
        if (p < N) { node_range(N, l, z[s], &na[s], &nbnd[s]); kb[p] = key[s] & 0xFFFFFFFFFFull; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      const int span = (N + nn - 1) / nn;          // the largest node of this level
      int rank[S];
      uint64_t mine[S];
#pragma unroll
      for (int s = 0; s < S; ++s) { rank[s] = 0; mine[s] = key[s] & 0xFFFFFFFFFFull; }
      for (int t = 0; t < span; ++t) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int j = na[s] + t;
          const uint64_t o = kb[j < nbnd[s] ? j : na[s]];
          rank[s] += (j < nbnd[s] && o < mine[s]) ? 1 : 0;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // every key has been read
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int p = lane + 64 * s;
        if (p < N) kb[na[s] + rank[s]] = key[s];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < S; ++s) { const int p = lane + 64 * s; key[s] = p < N ? kb[p] : ~0ull; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();   // kb is rewritten at the next level
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int p = lane + 64 * s;
      id[s] = p < N ? (int)(key[s] & 0xFFu) : 0;
#pragma unroll
      for (int d = 0; d < D; ++d) y[s][d] = ybuf[id[s] * D + d];
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int p = lane + 64 * s;
    if (p < N) {
      T->perm[p] = (uint8_t)id[s];
#pragma unroll
      for (int d = 0; d < D; ++d) T->ys[d][p] = y[s][d];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
  // ---- bottom-up statistics (n, mean, M2): level L slots z = lane + 64 s hold single points
  double h2[D];
#pragma unroll
  for (int d = 0; d < D; ++d) { const double h = fmax(hb[d], 1e-6); h2[d] = h * h; }
  int n[S];
  double m[S][D], M2[S][D];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int zz = lane + 64 * s;
    int a = 0, b = 0;
    if (zz < (1 << L)) node_range(N, L, zz, &a, &b);
    n[s] = b - a;
#pragma unroll
    for (int d = 0; d < D; ++d) { m[s][d] = n[s] > 0 ? (double)T->ys[d][a] : 0.0; M2[s][d] = 0.0; }
  }
  // (fast_rcp: v_rcp_f64 + two Newton steps; a level needs a dozen reciprocals.  The oracle's msg_build divides: the statistics agree to
  //  an ulp of double before they are rounded to single precision.)
  // parent node zz = lane + 64 ps of level l <- children 2 zz, 2 zz + 1 of level l + 1: child c lives in slot c >> 6, lane c & 63, i.e.
  // for parent slot ps in child slot 2 ps (lanes < 32) or 2 ps + 1 (kTwo: the child level has more than 64 nodes per pair of slots).
  // Same recurrences as msg_build.  NM = 128: one parent slot, the statements of the two-slot form it replaces.
  constexpr int PS = S / 2;
  auto merge_level = [&](int l, auto two_slots) {
#pragma clang fp contract(off)
    constexpr bool kTwo = decltype(two_slots)::value;
    int pn[PS]; double pm[PS][D], pM[PS][D], pinv[PS]; bool plive[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      const int zz = lane + 64 * ps;
      const int c0 = (2 * lane) & 127, sl = kTwo ? (c0 >> 6) : 0, la = c0 & 63, lb = (la + 1) & 63;
      constexpr int kLast = S - 1;
      const int ce = 2 * ps, co = 2 * ps + 1 <= kLast ? 2 * ps + 1 : kLast;
      int nl = __shfl(n[ce], la, 64), nr = __shfl(n[ce], lb, 64);
      if constexpr (kTwo) { const int nl1 = __shfl(n[co], la, 64), nr1 = __shfl(n[co], lb, 64); nl = sl ? nl1 : nl; nr = sl ? nr1 : nr; }
      plive[ps] = zz < (1 << l);
      const double nt = (double)(nl + nr);
      pinv[ps] = nl + nr > 0 ? fast_rcp(nt) : 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        double ml = shfl_f64(m[ce][d], la), mr = shfl_f64(m[ce][d], lb), Ml = shfl_f64(M2[ce][d], la), Mr = shfl_f64(M2[ce][d], lb);
        if constexpr (kTwo) {
          const double ml1 = shfl_f64(m[co][d], la), mr1 = shfl_f64(m[co][d], lb), Ml1 = shfl_f64(M2[co][d], la), Mr1 = shfl_f64(M2[co][d], lb);
          ml = sl ? ml1 : ml; mr = sl ? mr1 : mr; Ml = sl ? Ml1 : Ml; Mr = sl ? Mr1 : Mr;
        }
        double mm = 0.0, MM = 0.0;
        if (nl + nr > 0) {
          if (nr == 0) { mm = ml; MM = Ml; }
          else if (nl == 0) { mm = mr; MM = Mr; }
          else {
            const double dl = ml - mr;
            mm = ((double)nl * ml + (double)nr * mr) * pinv[ps];
            MM = Ml + Mr + (double)nl * (double)nr * pinv[ps] * dl * dl;
          }
        }
        pm[ps][d] = mm; pM[ps][d] = MM;
      }
      pn[ps] = plive[ps] ? nl + nr : 0;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) n[s] = 0;
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
      n[ps] = pn[ps];
#pragma unroll
      for (int d = 0; d < D; ++d) { m[ps][d] = pm[ps][d]; M2[ps][d] = pM[ps][d]; }
      if (plive[ps] && n[ps] > 0) {
        const int idn = (1 << l) - 1 + lane + 64 * ps;
        const double nt = (double)n[ps];
        double lg = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const double v = M2[ps][d] * pinv[ps] + h2[d];
          T->mean[d][idn] = (float)m[ps][d]; T->var[d][idn] = (float)v; T->ivar[d][idn] = (float)fast_rcp(v);
          lg += fast_log(v);   // (fdlibm kernel, < 1 ulp: the library call is twice the instructions)
        }
        T->cz[idn] = (float)(fast_log(nt * (1.0 / (double)N)) - 0.5 * lg);
      }
    }
  };
  if (L >= 1) merge_level(L - 1, std::true_type{});
  for (int l = L - 2; l >= 0; --l) {
    if constexpr (NM > 128) {   // (only NM = 256: a child level of 128 nodes sits in two slots; not even compiled into the 128 build)
      if ((2 << l) > 64) { merge_level(l, std::true_type{}); continue; }
    }
    merge_level(l, std::false_type{});
  }
  if (lane == 0) {
    double lg = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) { T->lvar[d] = (float)h2[d]; T->livar[d] = (float)(1.0 / h2[d]); lg += log(h2[d]); }
    T->lcz = (float)(log(1.0 / (double)N) - 0.5 * lg);
  }
}

template <int D, int NM>
__global__ void __launch_bounds__(256) k_gibbs_trees(const GibbsArgs a) {
  __shared__ float ybuf[4][NM * D];
  __shared__ int ext[4][(NM / 2) * 2 * D];
  __shared__ uint64_t kb[4][NM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + wave);
  if (row >= a.n_rows) return;   // wave-uniform; waves never synchronise with each other
  GibbsTree<D, NM>* T = reinterpret_cast<GibbsTree<D, NM>*>(a.trees) + row;
  gibbs_build<D, NM>(T, a.prop + (size_t)row * D * a.N, a.prop_bw + (size_t)row * D, row, a.N, a.L, a.circ, lane, ybuf[wave], ext[wave], kb[wave]);
}

// ---- candidate arithmetic: IEEE single precision, every operation spelled out (explicit fma, contraction off), so that the
// oracle (msg_exp32 / msg_ln32 / msg_wrap32 / msg_res_pair in oracle/rome_oracle.c) reproduces it bit for bit (an opt-in build takes exp and log from the
// hardware instead: ROME_GIBBS_HWTRANS below).  Written on
// f32x2 = two CANDIDATES of the same lane (nodes z, z + 1: their statistics are one 8-byte LDS broadcast), which the compiler maps to
// the packed v_pk_{add,mul,fma}_f32 instructions: half the VALU issue slots of the one-candidate-at-a-time form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr float kAbsent = -3.0e38f;   // log p of "no candidate" (odd N at the leaf level; also the initial running maximum)

__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }
// exp(max(x, -80)), x <= 0 (never 0: e^-80 = 1.8e-35 is below every acceptance threshold and every total); k = rint(x log2 e), Cody-Waite r = x - k ln2, Cephes expf polynomial, scaled by 2^k on the exponent field
// ROME_GIBBS_HWTRANS (default 0; 1 is an opt-in build): exp and log of the candidate arithmetic on the hardware transcendentals
// (v_exp_f32 / v_log_f32, within 1 ulp of the specified polynomials below, which the oracle evaluates: msg_exp32 / msg_ln32).  13 %
// faster (1.20 -> 1.02 ms per Manhattan sweep: the two polynomial exponentials are 65 of ~300 issue cycles of a candidate pair) and
// 50 784 of 50 784 product samples of the unit problems identical (scripts/gibbs_identical.py) -- but a categorical draw differs
// whenever a uniform lands within an ulp of a cumulative boundary, and over the 8e9 draws of two Manhattan-3500 solve iterations
// that happens: one of the 3500 pose means moved by 1.7e-3 against the oracle loop (north_star's bar is 1e-3).  Parity first: the
// shipped build evaluates the specified polynomials, bit for bit the oracle's.
#ifndef ROME_GIBBS_HWTRANS
#define ROME_GIBBS_HWTRANS 0
#endif
__device__ __forceinline__ f32x2 exp32_neg(f32x2 x) {
#pragma clang fp contract(off)
#if ROME_GIBBS_HWTRANS
  {
    const f32x2 t = __builtin_elementwise_max(x, splat(-80.0f)) * 1.44269504f;
    return f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  }
#endif
  const f32x2 xc = __builtin_elementwise_max(x, splat(-80.0f));
  const f32x2 k = __builtin_elementwise_rint(xc * 1.44269504f);
  f32x2 r = __builtin_elementwise_fma(k, splat(-0.693359375f), xc);
  r = __builtin_elementwise_fma(k, splat(2.12194440e-4f), r);
  f32x2 p = splat(1.9875691500e-4f);
  p = __builtin_elementwise_fma(p, r, splat(1.3981999507e-3f));
  p = __builtin_elementwise_fma(p, r, splat(8.3334519073e-3f));
  p = __builtin_elementwise_fma(p, r, splat(4.1665795894e-2f));
  p = __builtin_elementwise_fma(p, r, splat(1.6666665459e-1f));
  p = __builtin_elementwise_fma(p, r, splat(5.0000001201e-1f));
  const f32x2 y = __builtin_elementwise_fma(p, r * r, r) + 1.0f;
  const i32x2 ki = __builtin_convertvector(k, i32x2);
  const i32x2 bits = (i32x2)y + (ki << 23);
  return (f32x2)bits;
}
// the same function on ONE value in plain (unpacked) instructions: a packed operation issues at half the rate of a plain one on gfx950
// (profiles/r02_ubench_f32_issue.txt), so the rescale of the running total -- one value per lane -- through the packed form paid for a
// second, discarded element (round 6; the same IEEE operations in the same order: the same bits)
__device__ __forceinline__ float exp32_neg(float x) {
#pragma clang fp contract(off)
#if ROME_GIBBS_HWTRANS
  return __builtin_amdgcn_exp2f(fmaxf(x, -80.0f) * 1.44269504f);
#endif
  const float xc = fmaxf(x, -80.0f);
  const float k = __builtin_rintf(xc * 1.44269504f);
  float r = __builtin_fmaf(k, -0.693359375f, xc);
  r = __builtin_fmaf(k, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  const float y = __builtin_fmaf(p, r * r, r) + 1.0f;
  return __int_as_float(__float_as_int(y) + ((int)k << 23));
}
// ln(v), v > 0 normal: v = m 2^e, ln m the degree-7 polynomial in m - 1.5 of the Box-Muller radius (rome_device_math.hpp)
__device__ __forceinline__ f32x2 ln32_pos(f32x2 v) {
#pragma clang fp contract(off)
#if ROME_GIBBS_HWTRANS
  return f32x2{__builtin_amdgcn_logf(v.x), __builtin_amdgcn_logf(v.y)} * 0.693147181f;
#endif
  const u32x2 xb = (u32x2)v;
  const i32x2 e = (i32x2)(xb >> 23) - 127;
  const f32x2 ke = __builtin_convertvector(e, f32x2);
  const f32x2 t = (f32x2)((xb & 0x007FFFFFu) | 0x3F800000u) - 1.5f;
  f32x2 p = splat(0x1.4fab76p-7f);
  p = __builtin_elementwise_fma(p, t, splat(-0x1.1d4ffp-6f));
  p = __builtin_elementwise_fma(p, t, splat(0x1.a972ep-6f));
  p = __builtin_elementwise_fma(p, t, splat(-0x1.90d3ap-5f));
  p = __builtin_elementwise_fma(p, t, splat(0x1.94a6a8p-4f));
  p = __builtin_elementwise_fma(p, t, splat(-0x1.c72898p-3f));
  p = __builtin_elementwise_fma(p, t, splat(0x1.555544p-1f));
  p = __builtin_elementwise_fma(p, t, splat(0x1.9f324cp-2f));
  return __builtin_elementwise_fma(ke, splat(0x1.62e43p-1f), p);
}
// e - 2π rint(e / 2π) with 2π = 6.28125 + 1.9353072e-3 (the first part exact in 8 bits)
__device__ __forceinline__ f32x2 wrap32(f32x2 e) {
#pragma clang fp contract(off)
  const f32x2 k = __builtin_elementwise_rint(e * 0.15915494f);
  e = __builtin_elementwise_fma(k, splat(-6.28125f), e);
  return __builtin_elementwise_fma(k, splat(-1.9353072e-3f), e);
}
// one-pass categorical draw over candidate PAIRS (A, B): running maximum M of log p, total T of exp(log p - M); ONE uniform u per pair
// (top 24 bits of a xorshift32 stream, written as float(r | 255)·T_B against a·2^32): B replaces the selection when u T_B < a_B, A when
// a_B <= u T_B < a_A + a_B -- the probabilities of drawing after each candidate in turn, with half the generator steps (msg_res_pair)
struct Reservoir {
  float M, T; uint32_t r; int sel;
  __device__ __forceinline__ void init(uint32_t w) { M = kAbsent; T = 0.0f; r = w | 1u; sel = 0; }
  __device__ __forceinline__ void add2(int zA, int zB, f32x2 lp) {
#pragma clang fp contract(off)
    const float Mn = fmaxf(fmaxf(M, lp.x), lp.y);
    if (__builtin_amdgcn_ballot_w64(Mn > M) != 0) T = T * exp32_neg(M - Mn);   // exp32(0) = 1 exactly where M stays
    const f32x2 a = exp32_neg(lp - Mn);
    const float TA = T + a.x, TB = TA + a.y;
    r ^= r << 13; r ^= r >> 17; r ^= r << 5;
    const float lhs = (float)(r | 0xFFu) * TB;   // u in (0, 1]: 24 bits, never 0
    const f32x2 rhs = f32x2{a.x + a.y, a.y} * 4294967296.0f;
    sel = lhs < rhs.x ? zA : sel;
    sel = lhs < rhs.y ? zB : sel;
    T = TB; M = Mn;
  }
};
// Σ_d s_d / v_d and Π_d v_d of a group of GD <= 3 coordinates with ONE division: (Σ_d s_d Π_{e≠d} v_e) / Π_d v_d
template <int GD>
__device__ __forceinline__ void ratio_group(const f32x2* s, const f32x2* v, f32x2* q, f32x2* pv) {
#pragma clang fp contract(off)
  f32x2 num, den;
  if constexpr (GD == 3) {
    const f32x2 pab = v[0] * v[1];
    num = __builtin_elementwise_fma(s[0], v[1] * v[2], __builtin_elementwise_fma(s[1], v[0] * v[2], s[2] * pab));
    den = pab * v[2];
  } else {
    num = __builtin_elementwise_fma(s[0], v[1], s[1] * v[0]);
    den = v[0] * v[1];
  }
  *q = f32x2{num.x / den.x, num.y / den.y};   // (v_rcp_f32 + multiply instead of the IEEE division: no measurable gain, 1.040 vs 1.040 ms)
  *pv = den;
}

// Candidate scan with the statistics of the NEXT pair of candidates loaded while the current pair is evaluated: the LDS broadcasts of
// pair z + 2 are in flight during the ~80 dependent instructions of pair z (left to itself the compiler issues a pair's loads at the head
// of its own iteration and waits for them three instructions later).  Two register sets in ping-pong (no copies); the candidates are
// visited in the same order, so every draw is the one the plain loop makes.
#ifndef ROME_GIBBS_PREFETCH
#define ROME_GIBBS_PREFETCH 1   // 0: the plain loops (A/B on one box, scripts/gibbs_time.py: 2.545 -> 2.512 ms per bandwidth + tree + product pass,
#endif                          // i.e. -33 us = 2.8 % of the product kernel; asking for five waves per SIMD (94 VGPRs) on top: no gain)
template <class S, bool PREFETCH, class Load, class Body>
__device__ __forceinline__ void scan_pairs(int n, Load&& load, Body&& body) {
  if constexpr (!PREFETCH) {   // (Pose3: six coordinates -- the second register set would cost an occupancy step, 162 -> 170 VGPRs)
    for (int z = 0; z < n; z += 2) { S a; load(z, a); body(z, a); }
    return;
  }
  S a, b;
  load(0, a);
  for (int z = 0; z < n; z += 4) {
    load(z + 2 < n ? z + 2 : z, b);
    body(z, a);
    if (z + 2 >= n) break;
    load(z + 4 < n ? z + 4 : z + 2, a);
    body(z + 2, b);
  }
}
template <int D> struct PairIn { f32x2 m[D], v[D], c; };   // an inner level: means, (inverse) variances, log-weight constant of nodes z, z + 1
template <int D> struct PairLeaf { f32x2 y[D]; };           // the leaf level: the sorted single points p, p + 1

// LDS image of ONE level of one tree: what the candidate loops of that level read (wave-uniform -> LDS broadcasts, two nodes a read)
template <int D, int NM>
struct GibbsLevel {
  union {
    struct { float mean[D][NM / 2], var[D][NM / 2], ivar[D][NM / 2], cz[NM / 2], lnc[NM / 2]; } in;   // levels 1 .. L-1 (at most NM / 2 nodes); lnc = log(count / N)
    float ys[D][NM];                                                                                 // level L: the sorted single points
  };
};
template <int D>
struct GibbsConst { double ref[D], h[D], q0[4]; float lvar[D], livar[D]; int row; };

// CM: the circular mask at compile time (-1: read from the arguments) -- with a run-time mask the compiler evaluates the wrap of
// EVERY coordinate and selects (24 of the 54 instructions of a Pose2 candidate pair)
#ifdef ROME_GIBBS_TRACE   // experiment build (scripts/gibbs_phase_trace.py): where a lone block spends its time, phase by phase (block 0, thread 0; 10 ns ticks)
__device__ uint64_t g_gibbs_trace[16];
#define GTRACE(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const uint64_t now_ = wall_clock64(); g_gibbs_trace[slot] += now_ - tr_last; tr_last = now_; } } while (0)
#else
#define GTRACE(slot) do { } while (0)
#endif
template <int D, int CM, int NM>
__global__ void __launch_bounds__(NM) k_product_gibbs(const GibbsArgs a) {
#pragma clang fp contract(off)   // the candidate arithmetic below is a bit-exact specification
  constexpr int kGibbsThreads = NM;   // lane = output sample
  extern __shared__ __align__(16) unsigned char smem[];
#ifdef ROME_GIBBS_TRACE
  uint64_t tr_last = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int q = 0; q < 16; ++q) g_gibbs_trace[q] = 0;
#endif
  if ((int)blockIdx.x >= a.V) return;
  const int v = a.order[blockIdx.x];
  const int tid = threadIdx.x;
  const int N = a.N, L = a.L;
  const int k0 = a.prop_ptr[v], K = a.prop_ptr[v + 1] - k0;
  auto circ_bit = [&](int d) -> bool { if constexpr (CM >= 0) return (CM >> d) & 1; else return (a.circ >> d) & 1u; };
  const int vb = a.place.var_block ? a.place.var_block[v] : v;   // the block of the belief arrays this variable lives in
  double* ob = a.bel_out + (size_t)vb * D * N;
  const int msl = a.place.mirror_slot ? a.place.mirror_slot[v] : -1;
  double* mo = msl >= 0 ? a.place.mirror_out + (size_t)msl * (size_t)a.place.mirror_stride : nullptr;   // ALSO written: an exchange buffer
  // Arguments of the public entry that would index out of bounds -- more proposals than the caller's max_proposals sized the LDS
  // for, or a proposal row outside the tree workspace -- fail LOUDLY: the variable's belief becomes NaN (block-uniform test, no
  // out-of-bounds access); nothing is silently truncated.
  GTRACE(0);   // arguments
  bool bad = K < 0 || K > a.max_k;
  for (int j = 0; j < K && !bad; ++j) { const int r = a.prop_rows[k0 + j]; bad = r < 0 || r >= a.n_rows; }
  GTRACE(1);   // row check
  if (bad) {
    for (int q = tid; q < D * N; q += kGibbsThreads) { ob[q] = __builtin_nan(""); if (mo) mo[q] = __builtin_nan(""); }
    return;
  }
  if (K <= 1) {   // K = 0: the belief is kept; K = 1: the proposal is the product
    const double* src = K == 0 ? a.bel_in + (size_t)vb * D * N : a.prop + (size_t)a.prop_rows[k0] * D * N;
    for (int q = tid; q < D * N; q += kGibbsThreads) { const double x = src[q]; if (src != ob) ob[q] = x; if (mo) mo[q] = x; }
    return;
  }
  __shared__ float logn[NM + 1];   // log(c / N): the weight of a node with c points
  for (int c = tid; c <= NM; c += kGibbsThreads) logn[c] = c > 0 ? (float)log((double)c / (double)N) : 0.0f;
  // dynamic LDS: [k_lds] level images | [max_k] constants | [max_k][128] labels.  A variable with at most k_lds proposals keeps the
  // current level of all its trees resident; one with more STREAMS them: the image of the tree being scanned is staged into slot 0
  // before every categorical draw, and the statistics of selected nodes come from the trees in HBM / L2 (a handful of gathers per
  // draw).  Sizing every block for the hub variable of a pose graph (11 proposals: 28 kB) left 2.5 waves per SIMD; 12 kB give six
  // and the same work runs 24 % faster (profiles/r02_gibbs_occupancy.txt).
  GibbsLevel<D, NM>* lev = reinterpret_cast<GibbsLevel<D, NM>*>(smem);
  GibbsConst<D>* cst = reinterpret_cast<GibbsConst<D>*>(smem + sizeof(GibbsLevel<D, NM>) * a.k_lds);
  uint8_t* selbuf = smem + sizeof(GibbsLevel<D, NM>) * a.k_lds + sizeof(GibbsConst<D>) * a.max_k;
  const bool resident = K <= a.k_lds;
  const GibbsTree<D, NM>* __restrict__ W = reinterpret_cast<const GibbsTree<D, NM>*>(a.trees);
  for (int j = tid; j < K; j += kGibbsThreads) {
    const int row = a.prop_rows[k0 + j];
    const GibbsTree<D, NM>& T = W[row];
    GibbsConst<D>& c = cst[j];
#pragma unroll
    for (int d = 0; d < D; ++d) { c.ref[d] = T.ref[d]; c.h[d] = T.h[d]; c.lvar[d] = T.lvar[d]; c.livar[d] = T.livar[d]; }
    if constexpr (D == 6) { c.q0[0] = T.q0[0]; c.q0[1] = T.q0[1]; c.q0[2] = T.q0[2]; c.q0[3] = T.q0[3]; }
    c.row = row;
  }
  __syncthreads();
  // cooperative copy of level l of tree j into image slot `slot` (coalesced: the level's nodes are contiguous in each array of the tree)
  auto copy_level = [&](int l, int j, int slot) {
    const GibbsTree<D, NM>& T = W[cst[j].row];
    GibbsLevel<D, NM>& G = lev[slot];
    if (l < L) {
      const int n = 1 << l, o = n - 1;
      for (int q = tid; q < n * D; q += kGibbsThreads) {
        const int d = q >> l, z = q & (n - 1);
        G.in.mean[d][z] = T.mean[d][o + z]; G.in.var[d][z] = T.var[d][o + z]; G.in.ivar[d][z] = T.ivar[d][o + z];
      }
      for (int q = tid; q < n; q += kGibbsThreads) {
        int ra, rb; node_range(N, l, q, &ra, &rb);
        G.in.cz[q] = T.cz[o + q]; G.in.lnc[q] = logn[rb - ra];
      }
    } else {
      for (int q = tid; q < NM * D; q += kGibbsThreads) (&G.ys[0][0])[q] = (&T.ys[0][0])[q];
    }
  };
  auto stage = [&](int l) {   // resident variables: level l of every tree
    if (!resident) return;
    __syncthreads();   // everyone is done with the previous level's image
    for (int j = 0; j < K; ++j) copy_level(l, j, j);
    __syncthreads();
  };
  auto image = [&](int l, int j) -> const GibbsLevel<D, NM>& {   // the image the candidate loops of (level l, tree j) read
    if (resident) return lev[j];
    __syncthreads();
    copy_level(l, j, 0);
    __syncthreads();
    return lev[0];
  };
  // statistics of node sz of level l of tree i (a selected node: lane-divergent index): LDS image, or the tree itself when streaming
  auto node_mean = [&](int l, int i, int d, int sz) -> float {
    if (resident) return l < L ? lev[i].in.mean[d][sz] : lev[i].ys[d][sz];
    const GibbsTree<D, NM>& T = W[cst[i].row];
    return l < L ? T.mean[d][(1 << l) - 1 + sz] : T.ys[d][sz];
  };
  auto node_ivar = [&](int l, int i, int d, int sz) -> float {
    if (l == L) return cst[i].livar[d];
    return resident ? lev[i].in.ivar[d][sz] : W[cst[i].row].ivar[d][(1 << l) - 1 + sz];
  };
  // ---- sampling: lane = output sample
  const int s = tid;
  const bool live = s < N;
  const uint64_t st = a.stream_offset + (uint64_t)(a.place.var_stream ? a.place.var_stream[v] : v);
  uint32_t qu = 0, qn = 0;
  u32x4 wu = {0, 0, 0, 0}, wn = {0, 0, 0, 0};
  double npair0 = 0.0, npair1 = 0.0;
  auto uniform_word = [&]() -> uint32_t {
    if ((qu & 3u) == 0) wu = philox4x32_10(u32x4{(uint32_t)s, (uint32_t)st, (uint32_t)(st >> 32), (6u << 16) | (qu >> 2)}, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const uint32_t k = qu & 3u; ++qu;
    return k == 0 ? wu.x : (k == 1 ? wu.y : (k == 2 ? wu.z : wu.w));
  };
  auto normal = [&]() -> double {
    if ((qn & 3u) == 0) wn = philox4x32_10(u32x4{(uint32_t)s, (uint32_t)st, (uint32_t)(st >> 32), (7u << 16) | (qn >> 2)}, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    if ((qn & 1u) == 0) { if (qn & 2u) box_muller(wn.z, wn.w, &npair0, &npair1); else box_muller(wn.x, wn.y, &npair0, &npair1); }
    const double r = (qn & 1u) ? npair1 : npair0; ++qn;
    return r;
  };
  GTRACE(2);   // logn table, constants of the trees
  for (int j = 0; j < K; ++j) selbuf[j * NM + tid] = 0;
  stage(0);
  GTRACE(3);   // labels, level 0
  double x[D];
  [[maybe_unused]] double xq[4] = {1.0, 0.0, 0.0, 0.0};   // D = 6: rotation of the current point
  constexpr int DE = D == 6 ? 3 : D;                      // coordinates handled one by one (Euclidean / circular)
  for (int l = 1; l <= L + 1; ++l) {
    // (a) a point from the product of the selected nodes of level l - 1 (its image is the one in LDS).  Deviations are taken
    //     from density 0's node; rotations: Log(Q_0* ⊗ Q_j) with Q_j = q0_j ⊗ Exp(mean_ω) the node's absolute rotation.
    //     Labels of level L are sorted positions.
#pragma unroll
    for (int d = 0; d < DE; ++d) {
      double prec = 0.0, num = 0.0, mu0 = 0.0;
      for (int j = 0; j < K; ++j) {
        const GibbsConst<D>& c = cst[j];
        const int sz = selbuf[j * NM + tid];
        double mabs, iv;
        if (l - 1 == L) {   // the selected kernel itself: the particle at full precision, its bandwidth in double
          mabs = a.prop[(size_t)c.row * D * N + (size_t)d * N + W[c.row].perm[sz]];
          iv = 1.0 / (c.h[d] * c.h[d]);
        } else { mabs = c.ref[d] + (double)node_mean(l - 1, j, d, sz); iv = (double)node_ivar(l - 1, j, d, sz); }
        if (j == 0) mu0 = mabs;
        double dev = mabs - mu0;
        if (circ_bit(d)) dev = gwrap(dev);
        prec += iv; num += iv * dev;
      }
      const double xi = normal();
      x[d] = mu0 + num * fast_rcp(prec) + xi * fast_rsqrt(prec);   // (reciprocals by hardware seed + Newton: an ulp from the oracle's divisions)
    }
    if constexpr (D == 6) {
      double B[4] = {1.0, 0.0, 0.0, 0.0}, prec[3] = {0.0, 0.0, 0.0}, num[3] = {0.0, 0.0, 0.0};
      for (int j = 0; j < K; ++j) {
        const GibbsConst<D>& c = cst[j];
        const int sz = selbuf[j * NM + tid];
        double Q[4], iv[3];
        if (l - 1 == L) {
          const int pi = W[c.row].perm[sz];
          const double* pp = a.prop + (size_t)c.row * D * N + pi;
          const double w[3] = {pp[3 * (size_t)N], pp[4 * (size_t)N], pp[5 * (size_t)N]};
          quat_exp(w, Q);
#pragma unroll
          for (int k = 0; k < 3; ++k) iv[k] = 1.0 / (c.h[3 + k] * c.h[3 + k]);
        } else {
          const double m[3] = {(double)node_mean(l - 1, j, 3, sz), (double)node_mean(l - 1, j, 4, sz), (double)node_mean(l - 1, j, 5, sz)};
          double E[4]; quat_exp(m, E); quat_mul(c.q0, E, Q);
#pragma unroll
          for (int k = 0; k < 3; ++k) iv[k] = (double)node_ivar(l - 1, j, 3 + k, sz);
        }
        double dev[3] = {0.0, 0.0, 0.0};
        if (j == 0) { B[0] = Q[0]; B[1] = Q[1]; B[2] = Q[2]; B[3] = Q[3]; }
        else { double e[4]; quat_cmul(B, Q, e); quat_log(e, dev); }
#pragma unroll
        for (int k = 0; k < 3; ++k) { prec[k] += iv[k]; num[k] += iv[k] * dev[k]; }
      }
      double e[3], E[4];
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double xi = normal(); e[k] = num[k] * fast_rcp(prec[k]) + xi * fast_rsqrt(prec[k]); }
      quat_exp(e, E); quat_mul(B, E, xq);
    }
    GTRACE(4);   // (a) the point of level l - 1
    if (l == L + 1) break;
    stage(l);
    GTRACE(5);   // staging level l
    const int nz = 1 << l;
    // (c) labels of level l given the point, which is expressed ONCE in the density's own chart (Euclidean there) and rounded
    for (int j = 0; j < K; ++j) {
      const GibbsConst<D>& c = cst[j];
      const GibbsLevel<D, NM>& G = image(l, j);
      Reservoir R; R.init(uniform_word());
      float e0[D];
      {
        double e0d[D];
#pragma unroll
        for (int d = 0; d < DE; ++d) { e0d[d] = x[d] - c.ref[d]; if (circ_bit(d)) e0d[d] = gwrap(e0d[d]); }
        if constexpr (D == 6) { double e[4]; quat_cmul(c.q0, xq, e); quat_log(e, e0d + 3); }
#pragma unroll
        for (int d = 0; d < D; ++d) e0[d] = (float)e0d[d];
      }
      if (l < L) {   // wave-uniform candidates: node statistics are LDS broadcasts
        scan_pairs<PairIn<D>, (D <= 3 && ROME_GIBBS_PREFETCH)>(nz,
          [&](int z, PairIn<D>& P) {
#pragma unroll
            for (int d = 0; d < D; ++d) { P.m[d] = *reinterpret_cast<const f32x2*>(&G.in.mean[d][z]); P.v[d] = *reinterpret_cast<const f32x2*>(&G.in.ivar[d][z]); }
            P.c = *reinterpret_cast<const f32x2*>(&G.in.cz[z]);
          },
          [&](int z, const PairIn<D>& P) {
            f32x2 q = splat(0.0f);
#pragma unroll
            for (int d = 0; d < D; ++d) {
              f32x2 e = splat(e0[d]) - P.m[d];
              if (circ_bit(d)) e = wrap32(e);
              q = __builtin_elementwise_fma(e * e, P.v[d], q);
            }
            R.add2(z, z + 1, __builtin_elementwise_fma(splat(-0.5f), q, P.c));
          });
      } else {   // single points: every candidate has the kernel's variance and the weight 1/N: constants of the draw drop out
        scan_pairs<PairLeaf<D>, (D <= 3 && ROME_GIBBS_PREFETCH)>(N,
          [&](int p, PairLeaf<D>& P) {
#pragma unroll
            for (int d = 0; d < D; ++d) P.y[d] = *reinterpret_cast<const f32x2*>(&G.ys[d][p]);
          },
          [&](int p, const PairLeaf<D>& P) {
            f32x2 q = splat(0.0f);
#pragma unroll
            for (int d = 0; d < D; ++d) {
              f32x2 e = splat(e0[d]) - P.y[d];
              if (circ_bit(d)) e = wrap32(e);
              q = __builtin_elementwise_fma(e * e, splat(c.livar[d]), q);
            }
            f32x2 lp = q * -0.5f;
            if (p + 1 >= N) lp.y = kAbsent;
            R.add2(p, p + 1, lp);
          });
      }
      selbuf[j * NM + tid] = (uint8_t)R.sel;
    }
    GTRACE(6);   // (c) labels given the point
    // (d) Gibbs sweeps over the labels
    for (int it = 0; it < a.iters; ++it)
      for (int j = 0; j < K; ++j) {
        const GibbsConst<D>& c = cst[j];
        double Mx[D], Cx[D];   // product Gaussian of the other selected nodes, its mean in density j's chart
#pragma unroll
        for (int d = 0; d < DE; ++d) {
          double prec = 0.0, num = 0.0, mu0 = 0.0; bool first = true;
          for (int i = 0; i < K; ++i) {
            if (i == j) continue;
            const int sz = selbuf[i * NM + tid];
            double mean, iv;
            mean = (double)node_mean(l, i, d, sz); iv = (double)node_ivar(l, i, d, sz);
            const double mabs = cst[i].ref[d] + mean;
            if (first) { mu0 = mabs; first = false; }
            double dev = mabs - mu0;
            if (circ_bit(d)) dev = gwrap(dev);
            prec += iv; num += iv * dev;
          }
          const double ip = fast_rcp(prec);
          Mx[d] = mu0 + num * ip - c.ref[d]; Cx[d] = ip;
          if (circ_bit(d)) Mx[d] = gwrap(Mx[d]);
        }
        if constexpr (D == 6) {
          double B[4] = {1.0, 0.0, 0.0, 0.0}, prec[3] = {0.0, 0.0, 0.0}, num[3] = {0.0, 0.0, 0.0}; bool first = true;
          for (int i = 0; i < K; ++i) {
            if (i == j) continue;
            const int sz = selbuf[i * NM + tid];
            double m[3], iv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { m[k] = (double)node_mean(l, i, 3 + k, sz); iv[k] = (double)node_ivar(l, i, 3 + k, sz); }
            double E[4], Q[4], dev[3] = {0.0, 0.0, 0.0};
            quat_exp(m, E); quat_mul(cst[i].q0, E, Q);
            if (first) { B[0] = Q[0]; B[1] = Q[1]; B[2] = Q[2]; B[3] = Q[3]; first = false; }
            else { double e[4]; quat_cmul(B, Q, e); quat_log(e, dev); }
#pragma unroll
            for (int k = 0; k < 3; ++k) { prec[k] += iv[k]; num[k] += iv[k] * dev[k]; }
          }
          double e[3], E[4], QM[4], r[4];
#pragma unroll
          for (int k = 0; k < 3; ++k) { const double ip = fast_rcp(prec[k]); e[k] = num[k] * ip; Cx[3 + k] = ip; }
          quat_exp(e, E); quat_mul(B, E, QM); quat_cmul(c.q0, QM, r); quat_log(r, Mx + 3);
        }
        float mx[D], cx[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { mx[d] = (float)Mx[d]; cx[d] = (float)Cx[d]; }
        const GibbsLevel<D, NM>& G = image(l, j);
        Reservoir R; R.init(uniform_word());
        if (l < L) {
          scan_pairs<PairIn<D>, (D <= 3 && ROME_GIBBS_PREFETCH)>(nz,
            [&](int z, PairIn<D>& P) {
#pragma unroll
              for (int d = 0; d < D; ++d) { P.m[d] = *reinterpret_cast<const f32x2*>(&G.in.mean[d][z]); P.v[d] = *reinterpret_cast<const f32x2*>(&G.in.var[d][z]); }
              P.c = *reinterpret_cast<const f32x2*>(&G.in.lnc[z]);
            },
            [&](int z, const PairIn<D>& P) {
              f32x2 sq[D], vv[D];
#pragma unroll
              for (int d = 0; d < D; ++d) {
                f32x2 e = P.m[d] - splat(mx[d]);
                if (circ_bit(d)) e = wrap32(e);
                sq[d] = e * e;
                vv[d] = P.v[d] + splat(cx[d]);
              }
              // Σ_d e_d² / v_d + Σ_d log v_d with one division and one logarithm per group of <= 3 coordinates
              f32x2 q, pv;
              ratio_group<(D == 2 ? 2 : 3)>(sq, vv, &q, &pv);
              f32x2 t = q + ln32_pos(pv);
              if constexpr (D == 6) { ratio_group<3>(sq + 3, vv + 3, &q, &pv); t = t + (q + ln32_pos(pv)); }
              R.add2(z, z + 1, __builtin_elementwise_fma(splat(-0.5f), t, P.c));
            });
        } else {   // single points: every candidate has the variance h² + C and the weight 1/N
          float ivv[D];
#pragma unroll
          for (int d = 0; d < D; ++d) ivv[d] = 1.0f / (c.lvar[d] + cx[d]);
          scan_pairs<PairLeaf<D>, (D <= 3 && ROME_GIBBS_PREFETCH)>(N,
            [&](int p, PairLeaf<D>& P) {
#pragma unroll
              for (int d = 0; d < D; ++d) P.y[d] = *reinterpret_cast<const f32x2*>(&G.ys[d][p]);
            },
            [&](int p, const PairLeaf<D>& P) {
              f32x2 q = splat(0.0f);
#pragma unroll
              for (int d = 0; d < D; ++d) {
                f32x2 e = P.y[d] - splat(mx[d]);
                if (circ_bit(d)) e = wrap32(e);
                q = __builtin_elementwise_fma(e * e, splat(ivv[d]), q);
              }
              f32x2 lp = q * -0.5f;
              if (p + 1 >= N) lp.y = kAbsent;
              R.add2(p, p + 1, lp);
            });
        }
        selbuf[j * NM + tid] = (uint8_t)R.sel;
      }
    GTRACE(7);   // (d) Gibbs sweep
  }
  if constexpr (D == 6) quat_log(xq, x + 3);
  if (live) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double xo = (circ_bit(d)) ? gwrap(x[d]) : x[d];
      ob[(size_t)d * N + s] = xo;
      if (mo) mo[(size_t)d * N + s] = xo;
    }
  }
}

#ifdef ROME_GIBBS_TRACE
}  // namespace rome
extern "C" int rome_debug_gibbs_trace(unsigned long long* out16) {
  return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(rome::g_gibbs_trace), sizeof(unsigned long long) * 16);
}
namespace rome {
#endif
static size_t tree_bytes(int dim, int n_rows, int N) {
  const size_t one = N <= 128 ? (dim == 2 ? sizeof(GibbsTree<2, 128>) : (dim == 3 ? sizeof(GibbsTree<3, 128>) : sizeof(GibbsTree<6, 128>)))
                              : (dim == 2 ? sizeof(GibbsTree<2, 256>) : (dim == 3 ? sizeof(GibbsTree<3, 256>) : sizeof(GibbsTree<6, 256>)));
  return one * (size_t)(n_rows > 0 ? n_rows : 1);
}
// workspace: one tree per proposal row | the variables ordered by their number of proposals
size_t gibbs_workspace_bytes(int dim, int n_rows, int V, int N) { return tree_bytes(dim, n_rows, N) + sizeof(int32_t) * (size_t)(V > 0 ? V : 1); }

// Variables in descending order of their number of proposals (counting sort, one block): the blocks of the sampling kernel are
// dispatched in this order, i.e. longest first (a block's run time is proportional to its variable's proposal count, 2 .. 11 on
// Manhattan), and the short ones fill in behind them -- list scheduling by the hardware dispatcher.  In graph order the hub's block
// started wherever its index fell and the launch ended with a few SIMDs still busy (1.9 -> 1.25 ms together with the rest).  The
// order inside a count is whatever the atomics give -- the product of a variable does not depend on it.
__global__ void __launch_bounds__(1024) k_gibbs_order(int V, const int32_t* __restrict__ prop_ptr, int32_t* __restrict__ order) {
  __shared__ int hist[65], start[65];
  for (int k = threadIdx.x; k < 65; k += 1024) hist[k] = 0;
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += 1024) { const int K = prop_ptr[v + 1] - prop_ptr[v]; atomicAdd(&hist[K < 64 ? K : 64], 1); }
  __syncthreads();
  if (threadIdx.x == 0) { int acc = 0; for (int k = 64; k >= 0; --k) { start[k] = acc; acc += hist[k]; } }
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += 1024) { const int K = prop_ptr[v + 1] - prop_ptr[v]; order[atomicAdd(&start[K < 64 ? K : 64], 1)] = v; }
}

template <int NM>
static hipError_t launch_product_gibbs_nm(GibbsArgs& a, int dim, int V, int N, int n_rows, uint32_t circ, int max_k, hipStream_t s) {
  int32_t* order = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(a.trees) + tree_bytes(dim, n_rows, N));
  a.order = order;
  hipLaunchKernelGGL(k_gibbs_order, dim3(1), dim3(1024), 0, s, V, a.prop_ptr, order);
  if (n_rows > 0 && max_k > 1) {   // (no variable has two proposals: every product is a copy, no tree is read)
    if (dim == 2) hipLaunchKernelGGL((k_gibbs_trees<2, NM>), dim3((n_rows + 3) / 4), dim3(256), 0, s, a);
    else if (dim == 3) hipLaunchKernelGGL((k_gibbs_trees<3, NM>), dim3((n_rows + 3) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_gibbs_trees<6, NM>), dim3((n_rows + 3) / 4), dim3(256), 0, s, a);
  }
  // LDS of a block: kGibbsResident level images (variables with more proposals stream theirs through slot 0) + constants and labels
  // for the variable with the most proposals
  constexpr int kGibbsResident = 4;
  const size_t img = dim == 2 ? sizeof(GibbsLevel<2, NM>) : (dim == 3 ? sizeof(GibbsLevel<3, NM>) : sizeof(GibbsLevel<6, NM>));
  const size_t per = (dim == 2 ? sizeof(GibbsConst<2>) : (dim == 3 ? sizeof(GibbsConst<3>) : sizeof(GibbsConst<6>))) + NM;
  a.k_lds = max_k < kGibbsResident ? max_k : kGibbsResident;
  // A launch that cannot fill the chip anyway (a clique, a small frontier: at most two blocks per CU) has no occupancy to protect: every
  // level image resident, as far as 64 kB of LDS go -- the streaming path re-stages an image before every categorical draw, and a
  // block's latency is all such a launch pays (same draws either way: the images hold the same numbers).
  if (V <= 512 && max_k > kGibbsResident) {
    const size_t room = 64 * 1024 - per * (size_t)max_k;
    const int fit = (int)(room / img);
    a.k_lds = max_k < fit ? max_k : (fit > kGibbsResident ? fit : kGibbsResident);
  }
  const size_t bytes = img * (size_t)a.k_lds + per * (size_t)max_k;
  if (bytes > 150 * 1024) return hipErrorInvalidValue;
  auto launch = [&](auto kernel) -> hipError_t {
    if (bytes > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(V), dim3(NM), bytes, s, a);
    return hipGetLastError();
  };
  if (dim == 2) return circ == 0 ? launch(k_product_gibbs<2, 0, NM>) : launch(k_product_gibbs<2, -1, NM>);
  if (dim == 3) return circ == 4u ? launch(k_product_gibbs<3, 4, NM>) : (circ == 0 ? launch(k_product_gibbs<3, 0, NM>) : launch(k_product_gibbs<3, -1, NM>));
  return circ == 0 ? launch(k_product_gibbs<6, 0, NM>) : launch(k_product_gibbs<6, -1, NM>);
}

hipError_t launch_product_gibbs(int dim, int V, int N, int n_rows, const int32_t* prop_ptr, const int32_t* prop_rows, const double* prop,
                                const double* prop_bw, const double* bel_in, double* bel_out, void* trees, uint32_t circ, int iters, int max_k,
                                uint64_t seed, uint64_t stream_offset, hipStream_t s, const GibbsPlace* place) {
  if (V <= 0) return hipSuccess;
  if (N < 1 || N > kGibbsMaxN || (dim != 2 && dim != 3 && dim != 6) || max_k < 1 || n_rows < 0) return hipErrorInvalidValue;
  GibbsArgs a;
  a.V = V; a.N = N; a.L = 0; while ((1 << a.L) < N) ++a.L;
  a.max_k = max_k; a.iters = iters < 1 ? 1 : iters; a.circ = circ;
  a.prop_ptr = prop_ptr; a.prop_rows = prop_rows; a.prop = prop; a.prop_bw = prop_bw; a.bel_in = bel_in; a.bel_out = bel_out;
  a.trees = trees; a.n_rows = n_rows;
  a.seed = seed; a.stream_offset = stream_offset;
  a.place = place ? *place : GibbsPlace{nullptr, nullptr, nullptr, nullptr, 0};
  if (!a.place.mirror_out) a.place.mirror_slot = nullptr;
  // N <= 128: the 128-thread / two-slot instantiation (what every measured number of the solve loop is quoted on); above: 256
  return N <= 128 ? launch_product_gibbs_nm<128>(a, dim, V, N, n_rows, circ, max_k, s) : launch_product_gibbs_nm<256>(a, dim, V, N, n_rows, circ, max_k, s);
}

}  // namespace rome
