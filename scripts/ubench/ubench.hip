// micro-benchmarks of the per-lane building blocks (wave-instruction cost on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../rome.jl_amd/csrc/rome_device_math.hpp"
using namespace rome;

#define ITERS 256
__device__ __forceinline__ u32x4 philox_mad64(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
    c = u32x4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b) {
  uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc"); return r;
}
__device__ __forceinline__ u32x4 philox_asm(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = mad64(0xD2511F53u, c.x), p1 = mad64(0xCD9E8D57u, c.z);
    c = u32x4{(uint32_t)(p1 >> 32) ^ c.y ^ k0, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k1, (uint32_t)p0};
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c;
}
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
    if constexpr (WHICH == 13) { u32x4 w = philox_mad64(u32x4{ctr++, 1u, 2u, 3u}, 5u, 6u); acc += (double)(w.x ^ w.y ^ w.z ^ w.w); }
    if constexpr (WHICH == 14) { u32x4 w = philox_asm(u32x4{ctr++, 1u, 2u, 3u}, 5u, 6u); acc += (double)(w.x ^ w.y ^ w.z ^ w.w); }
    if constexpr (WHICH == 15) { double s, c; fast_sincos(x, &s, &c); acc += s * c; x += 0.37; }
    if constexpr (WHICH == 16) { acc += fast_log(x * x + 1.0); x += 0.37; }
    if constexpr (WHICH == 17) { acc += fast_sqrt(x * x + 1.0); x += 0.37; }
    if constexpr (WHICH == 18) { double v[6] = {x, x + 1, x + 2, x + 3, x + 4, x + 5}; wave_sum_n<6>(v); acc += v[0] + v[5]; x += 0.37; }
    if constexpr (WHICH == 19) { acc += wrap_pi(x); x += 0.37; }
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
  const char* names[] = {"philox4x32_10", "sincos(ocml)", "atan2", "log", "sqrt", "wave_sum(shfl_xor x6)", "fma chain", "sincospi", "remainder", "wrap_pi(sincos+atan2)", "rng_normals<3>", "acos", "fdiv", "philox (u64 product)", "philox (v_mad_u64_u32 asm)", "fast_sincos", "fast_log", "fast_sqrt", "wave_sum_n<6> (DPP)", "wrap_pi (fast)"};
  float ms[20];
  ms[0] = run<0>(d, blocks); ms[1] = run<1>(d, blocks); ms[2] = run<2>(d, blocks); ms[3] = run<3>(d, blocks); ms[4] = run<4>(d, blocks);
  ms[5] = run<5>(d, blocks); ms[6] = run<6>(d, blocks); ms[7] = run<7>(d, blocks); ms[8] = run<8>(d, blocks); ms[9] = run<9>(d, blocks);
  ms[10] = run<10>(d, blocks); ms[11] = run<11>(d, blocks); ms[12] = run<12>(d, blocks);
  ms[13] = run<13>(d, blocks); ms[14] = run<14>(d, blocks); ms[15] = run<15>(d, blocks); ms[16] = run<16>(d, blocks);
  ms[17] = run<17>(d, blocks); ms[18] = run<18>(d, blocks); ms[19] = run<19>(d, blocks);
  // waves = blocks*4 ; per-SIMD waves = waves/1024 ; cycles per call per wave = ms*2.4e6 / (ITERS * waves/1024)
  const double waves_per_simd = blocks * 4.0 / 1024.0;
  for (int i = 0; i < 20; ++i)
    printf("%-26s %8.3f ms  -> %7.1f SIMD-cycles per wave-call (@2.4GHz)\n", names[i], ms[i], ms[i] * 2.4e6 / (ITERS * waves_per_simd));
  return 0;
}
