// micro-benchmarks of the per-lane building blocks (wave-instruction cost on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../rome.jl_amd/csrc/rome_device_math.hpp"
using namespace rome;

#define ITERS 256
template <int WHICH>
__global__ void __launch_bounds__(256) k(double* out, double seed) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  double x = seed + 1e-3 * (tid & 1023), acc = 0;
  uint32_t ctr = tid;
  for (int i = 0; i < ITERS; ++i) {
    if constexpr (WHICH == 0) { u32x4 w = philox4x32_10(u32x4{ctr++, 1u, 2u, 3u}, 5u, 6u); acc += (double)(w.x ^ w.y ^ w.z ^ w.w); }
    if constexpr (WHICH == 1) { double s, c; sincos(x, &s, &c); acc += s * c; x += 0.37; }
    if constexpr (WHICH == 2) { acc += atan2(x, acc + 1.0); x += 0.37; }
    if constexpr (WHICH == 3) { acc += log(x * x + 1.0); x += 0.37; }
    if constexpr (WHICH == 4) { acc += sqrt(x * x + 1.0); x += 0.37; }
    if constexpr (WHICH == 5) { acc += wave_sum(x); x += 0.37; }
    if constexpr (WHICH == 6) { acc = fma(acc, 1.0000001, x); }  // dependent FMA chain
    if constexpr (WHICH == 7) { double s, c; sincospi(x, &s, &c); acc += s * c; x += 0.37; }
    if constexpr (WHICH == 8) { acc += remainder(x, 6.283185307179586); x += 0.37; }
    if constexpr (WHICH == 9) { acc += wrap_pi(x); x += 0.37; }
    if constexpr (WHICH == 10) { double n[3]; rng_normals<3>(7, 9, ctr++, n); acc += n[0] + n[1] + n[2]; }
    if constexpr (WHICH == 11) { acc += acos(fmin(1.0, fabs(x) * 1e-3)); x += 0.37; }
    if constexpr (WHICH == 12) { acc += x / (acc + 2.0); x += 0.37; }
  }
  out[tid] = acc;
}
template <int W> float run(double* d, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(256), 0, 0, d, 0.5);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<W>), dim3(blocks), dim3(256), 0, 0, d, 0.5);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  const int blocks = 256 * 8 * 4;  // 8 waves/SIMD x 4 rounds
  double* d; hipMalloc(&d, sizeof(double) * blocks * 256);
  const char* names[] = {"philox4x32_10", "sincos(ocml)", "atan2", "log", "sqrt", "wave_sum(shfl_xor x6)", "fma chain", "sincospi", "remainder", "wrap_pi(sincos+atan2)", "rng_normals<3>", "acos", "fdiv"};
  float ms[13];
  ms[0] = run<0>(d, blocks); ms[1] = run<1>(d, blocks); ms[2] = run<2>(d, blocks); ms[3] = run<3>(d, blocks); ms[4] = run<4>(d, blocks);
  ms[5] = run<5>(d, blocks); ms[6] = run<6>(d, blocks); ms[7] = run<7>(d, blocks); ms[8] = run<8>(d, blocks); ms[9] = run<9>(d, blocks);
  ms[10] = run<10>(d, blocks); ms[11] = run<11>(d, blocks); ms[12] = run<12>(d, blocks);
  // waves = blocks*4 ; per-SIMD waves = waves/1024 ; cycles per call per wave = ms*2.4e6 / (ITERS * waves/1024)
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  for (int i = 0; i < 13; ++i)
    printf("%-26s %8.3f ms  -> %7.1f SIMD-cycles per wave-call (@2.4GHz)\n", names[i], ms[i], ms[i] * 2.4e6 / (ITERS * waves_per_simd));
  return 0;
}
