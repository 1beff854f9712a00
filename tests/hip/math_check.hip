// Unit check of the device math in rome.jl_amd/csrc/rome_device_math.hpp: evaluates every tuned elementary function on the points
// of an input file and writes the results; tests/test_gpu_device_math.py compiles this with hipcc, runs it and compares with numpy.
//   math_check <in.bin> <out.bin>     in: n doubles x, n doubles y     out: 9 arrays of n doubles
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../rome.jl_amd/csrc/rome_device_math.hpp"
using namespace rome;

__global__ void k(int n, const double* x, const double* y, double* o) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c; fast_sincos(x[i], &s, &c);
  o[0 * n + i] = s; o[1 * n + i] = c;
  o[2 * n + i] = wrap_pi(x[i]);
  o[3 * n + i] = fast_sqrt(fabs(x[i]));
  o[4 * n + i] = fast_log(fabs(y[i]) + 1e-300);
  o[5 * n + i] = fast_exp_neg(-fabs(x[i]));
  o[6 * n + i] = fast_atan2(y[i], x[i]);
  double w[3] = {x[i] * 0.01, y[i] * 0.01, (x[i] - y[i]) * 0.01}, q[4], back[3];
  quat_exp(w, q); quat_log(q, back);
  o[7 * n + i] = back[0] - w[0]; o[8 * n + i] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] - 1.0;
}

// wave64 reductions: every lane contributes 4 (and 3) values; all lanes must end with the totals
__global__ void k_wave(const double* x, double* o) {
  const int l = threadIdx.x;
  double v4[4] = {x[l], x[64 + l], x[128 + l], x[192 + l]};
  double v3[3] = {x[l], x[64 + l], x[128 + l]};
  wave_sum_n<4>(v4); wave_sum_n<3>(v3);
  for (int k = 0; k < 4; ++k) o[k * 64 + l] = v4[k];
  for (int k = 0; k < 3; ++k) o[256 + k * 64 + l] = v3[k];
}
static int check_wave(const double* dx) {
  double* dw; if (hipMalloc(&dw, 8 * 448) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_wave, dim3(1), dim3(64), 0, 0, dx, dw);
  std::vector<double> w(448), in(256);
  if (hipMemcpy(w.data(), dw, 8 * 448, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (hipMemcpy(in.data(), dx, 8 * 256, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int k = 0; k < 4; ++k) {
    long double ref = 0, mag = 0;
    for (int l = 0; l < 64; ++l) { ref += in[k * 64 + l]; mag += fabsl(in[k * 64 + l]); }
    for (int l = 0; l < 64; ++l) {
      if (fabsl(w[k * 64 + l] - ref) > 1e-14L * mag + 1e-300L) { printf("wave_sum_n<4> value %d lane %d: %g vs %Lg\n", k, l, w[k * 64 + l], ref); return 1; }
      if (w[k * 64 + l] != w[k * 64]) { printf("wave_sum_n<4>: lanes disagree\n"); return 1; }
      if (k < 3 && fabsl(w[256 + k * 64 + l] - ref) > 1e-14L * mag + 1e-300L) { printf("wave_sum_n<3> value %d lane %d\n", k, l); return 1; }
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
  const int n = (int)(bytes / 16);
  std::vector<double> h(2 * (size_t)n);
  if (fread(h.data(), 8, 2 * (size_t)n, f) != 2 * (size_t)n) return 4;
  fclose(f);
  double *dx, *dout;
  if (hipMalloc(&dx, 16 * (size_t)n) != hipSuccess || hipMalloc(&dout, 72 * (size_t)n) != hipSuccess) return 5;
  if (hipMemcpy(dx, h.data(), 16 * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return 7;
  if (n >= 256 && check_wave(dx)) return 8;
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, n, dx, dx + n, dout);
  std::vector<double> out(9 * (size_t)n);
  if (hipMemcpy(out.data(), dout, 72 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return 6;
  f = fopen(argv[2], "wb"); fwrite(out.data(), 8, out.size(), f); fclose(f);
  printf("math_check ok %d\n", n);
  return 0;
}
