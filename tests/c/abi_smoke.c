/* Plain-C caller of librome_mi355.so -- what a non-Python host (the Julia ccall shim) sees.
 * Build: gcc tests/c/abi_smoke.c -Iinclude -Lrome.jl_amd -lrome_mi355 -lm -o /tmp/abi_smoke
 * Checks, with no oracle: (1) reference known answers of the Pose2Pose2 residual functor
 * (test/testParametricSimulated.jl:42-46), (2) a Pose2Pose2 convolution with pre-sampled noise against the
 * closed-form root of SURVEY Appendix A.5 computed right here, (3) error codes. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "rome_mi355.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != ROME_OK) { printf("FAIL %s -> %d (%s)\n", #x, rc_, rome_strerror(rc_)); return 1; } } while (0)

int main(void) {
  rome_ctx* ctx = NULL;
  int rc = rome_ctx_create(&ctx, 0);
  if (rc == ROME_ERR_NO_DEVICE) { printf("no device: %s\n", rome_strerror(rc)); return 77; }
  CHECK(rc);
  const double PI = 3.14159265358979323846;
  /* (1) z = (0,0,-pi), p = identity, q = (0,0,±pi) -> r = 0 */
  double z[6] = {0, 0, -PI, 0, 0, -PI}, p[6] = {0, 0, 0, 0, 0, 0}, q[6] = {0, 0, -PI, 0, 0, PI}, r[6];
  CHECK(rome_residual_pose2pose2(ctx, 2, z, p, q, r));
  for (int k = 0; k < 6; ++k) if (fabs(r[k]) > 1e-14) { printf("FAIL residual KAT r[%d]=%g\n", k, r[k]); return 1; }
  /* (2) one convolution, N = 100, dir 0, AoS layout, noise given */
  enum { N = 100 };
  rome_opts o; rome_opts_default(&o, ROME_SOLVER_NEWTON);
  o.n_particles = N; o.layout = ROME_LAYOUT_AOS;
  double mu[3] = {10.0, 0.0, PI / 3}, cov[9] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
  double fixed[N * 3], noise[N * 3], target[N * 3];
  int32_t status[N], dir[1] = {0};
  for (int i = 0; i < N; ++i) {
    fixed[3 * i] = 0.01 * i; fixed[3 * i + 1] = -0.02 * i; fixed[3 * i + 2] = 0.03 * i - 1.0;
    noise[3 * i] = sin(1.0 * i); noise[3 * i + 1] = cos(2.0 * i); noise[3 * i + 2] = sin(3.0 * i + 1);
    target[3 * i] = 5; target[3 * i + 1] = 5; target[3 * i + 2] = 0.5;
  }
  CHECK(rome_conv_pose2pose2(ctx, &o, 1, dir, mu, cov, fixed, noise, target, status));
  double worst = 0;
  for (int i = 0; i < N; ++i) {
    const double zx = mu[0] + 0.1 * noise[3 * i], zy = mu[1] + 0.1 * noise[3 * i + 1], zt = mu[2] + 0.1 * noise[3 * i + 2];
    const double th = fixed[3 * i + 2], c = cos(th), s = sin(th);
    const double ex = fixed[3 * i] + c * zx - s * zy, ey = fixed[3 * i + 1] + s * zx + c * zy;
    const double dt = remainder(th + zt - target[3 * i + 2], 2 * PI);
    worst = fmax(worst, fmax(fabs(ex - target[3 * i]), fmax(fabs(ey - target[3 * i + 1]), fabs(dt))));
    if (status[i] != 0) { printf("FAIL status[%d]=%d\n", i, status[i]); return 1; }
  }
  if (worst > 1e-10) { printf("FAIL convolution vs closed form: %g\n", worst); return 1; }
  /* (3) error behaviour */
  cov[0] = -1.0;
  if (rome_conv_pose2pose2(ctx, &o, 1, dir, mu, cov, fixed, noise, target, status) != ROME_ERR_NOT_POSDEF) { printf("FAIL posdef\n"); return 1; }
  o.n_particles = ROME_MAX_PARTICLES + 1;
  if (rome_conv_pose2pose2(ctx, &o, 1, dir, mu, cov, fixed, noise, target, status) != ROME_ERR_UNSUPPORTED_N) { printf("FAIL maxN\n"); return 1; }
  if (rome_conv_pose2pose2(NULL, &o, 1, dir, mu, cov, fixed, noise, target, status) != ROME_ERR_UNSUPPORTED_N &&
      rome_conv_pose2pose2(NULL, &o, 1, dir, mu, cov, fixed, noise, target, status) != ROME_ERR_INVALID_ARG) { printf("FAIL null ctx\n"); return 1; }
  rome_ctx_destroy(ctx);
  printf("abi_smoke ok (max |Δ| vs closed form %.2e)\n", worst);
  return 0;
}
