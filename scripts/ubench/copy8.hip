// HBM-counter calibration: a plain 8-byte-per-lane streaming copy of a known size (>> Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}
int main() {
  const size_t n = (size_t)1 << 28;  // 2 GiB read + 2 GiB written
  double *a, *b; hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 1, n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy8, dim3(256 * 8 * 4), dim3(256), 0, 0, a, b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("copy8: %zu bytes read + %zu written in %.3f ms -> %.1f GB/s\n", n * 8, n * 8, ms, 2.0 * n * 8 / ms / 1e6);
  }
  return 0;
}
