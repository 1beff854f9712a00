// issue cost of single-precision building blocks on gfx950 (8 waves/SIMD resident): SIMD-cycles per wave-instruction
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define ITERS 512
template <int WHICH>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[8];
  f32x2 p[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { a[u] = seed * (float)(tid & 63) * 1e-3f - (float)u; p[u] = f32x2{a[u], a[u] * 0.5f}; }
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (WHICH == 0) a[u] = __builtin_amdgcn_exp2f(a[u]) - 1.5f;                       // exp + add
      if constexpr (WHICH == 1) a[u] = __builtin_fmaf(a[u], 0.999f, -0.5f);                       // fma
      if constexpr (WHICH == 2) p[u] = __builtin_elementwise_fma(p[u], f32x2{0.999f, 0.998f}, f32x2{-0.5f, -0.25f});   // pk_fma
      if constexpr (WHICH == 3) a[u] = __builtin_fmaf(a[u], 0.999f, -0.5f) + 0.25f;               // fma + add
      if constexpr (WHICH == 4) a[u] = __builtin_amdgcn_rcpf(a[u]) - 1.5f;                        // rcp + add
      if constexpr (WHICH == 5) a[u] = __builtin_rintf(a[u] * 1.01f) - 0.5f;                      // mul, rndne, add
      if constexpr (WHICH == 6) { p[u].x = __builtin_amdgcn_exp2f(p[u].x); p[u].y = __builtin_amdgcn_exp2f(p[u].y); p[u] = p[u] - 1.5f; }   // 2 exp + pk_add
    }
  }
  float acc = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) acc += a[u] + p[u].x + p[u].y;
  out[tid] = acc;
}
template <int W> float run(float* d, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(256), 0, 0, d, 1.0f); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(256), 0, 0, d, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  const int blocks = 256 * 8 * 4;
  float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
  const char* names[] = {"v_exp_f32 + v_add", "v_fma_f32", "v_pk_fma_f32", "v_fma + v_add", "v_rcp_f32 + v_add", "v_mul + v_rndne + v_add", "2 v_exp + v_pk_add"};
  const int ninstr[] = {2, 1, 1, 2, 2, 3, 3};
  float ms[7] = {run<0>(d, blocks), run<1>(d, blocks), run<2>(d, blocks), run<3>(d, blocks), run<4>(d, blocks), run<5>(d, blocks), run<6>(d, blocks)};
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  for (int i = 0; i < 7; ++i) {
    const double cyc = ms[i] * 2.4e6 / (ITERS * 8.0 * waves_per_simd);
    printf("%-26s %8.3f ms -> %6.2f SIMD-cycles per group of %d instructions\n", names[i], ms[i], cyc, ninstr[i]);
  }
  return 0;
}
