/* Plain-C caller of librome_mi355.so -- what a non-Python host (the Julia ccall shim) sees.
 * Build: gcc tests/c/abi_smoke.c -Iinclude -Lrome.jl_amd -lrome_mi355 -lm -o /tmp/abi_smoke
 * Checks, with no oracle: (1) reference known answers of the Pose2Pose2 residual functor
 * (test/testParametricSimulated.jl:42-46), (2) a Pose2Pose2 convolution with pre-sampled noise against the
 * closed-form root of SURVEY Appendix A.5 computed right here, (3) error codes, (4) a clique up-solve (IIF upGibbsCliqueDensity) in one
 * call: x1 with an odometry factor from x0 and its own prior -> the product sits where both agree, (5) a `multihypo=[1,.5,.5]`
 * bearing-range row in the clique entry (rome_clique_host.br1_alt / br1_hypo_w): the pose proposals land on the rings around BOTH
 * landmark candidates, (6) the device-resident form: rome_store + rome_upsolve_plan + rome_scatter_plan reproduce (4) bit for bit
 * with the beliefs staying on the device. */
#include <string.h>
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
  /* (4) rome_clique_upsolve: variables x0 (index 0), x1 (index 1); rows targeting x1: odometry x0 -> x1 (dir 0) and PriorPose2(x1) */
  {
    static double bel[2 * 3 * N], nb[3 * N], bw[3];
    rome_opts ou; rome_opts_default(&ou, ROME_SOLVER_NEWTON); ou.n_particles = N; ou.layout = ROME_LAYOUT_SOA; ou.seed = 5;
    for (int i = 0; i < N; ++i) {
      bel[0 * N + i] = 0.05 * sin(7.0 * i); bel[1 * N + i] = 0.05 * cos(3.0 * i); bel[2 * N + i] = 0.01 * sin(1.3 * i);   /* x0 around the origin */
      bel[3 * N + 0 * N + i] = 3.0 + sin(0.7 * i); bel[3 * N + 1 * N + i] = cos(0.9 * i); bel[3 * N + 2 * N + i] = 0.3 * sin(2.1 * i);  /* x1: poor start */
    }
    const int32_t rows[8] = {0, 0, 0, 1,   /* factor 0, dir 0 (solve the 2nd variable), fixed x0, target x1 */
                             1, 2, 1, 1};  /* factor 1, prior row on x1 */
    const double mu2[6] = {10.0, 0.0, 0.0, 10.0, 0.0, 0.0};
    const double cv2[18] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.0001,   0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.0001};
    const int32_t upt[1] = {0}, upv[1] = {1};
    rome_clique_upsolve_host u; memset(&u, 0, sizeof(u));
    u.clique.n_pose2 = 2; u.clique.bel_pose2 = bel;
    u.clique.n_p2p2 = 2; u.clique.f_p2p2 = 2; u.clique.p2p2_rows4 = rows; u.clique.p2p2_mu = mu2; u.clique.p2p2_cov = cv2;
    u.gibbs_iters = 3; u.product_iters = 1; u.schedule = ROME_UPSOLVE_SEQUENTIAL; u.n_up = 1; u.up_type = upt; u.up_var = upv;
    u.new_pose2 = nb; u.bw_pose2 = bw;
    CHECK(rome_clique_upsolve(ctx, &ou, &u));
    double mx = 0, my = 0, mt = 0;
    for (int i = 0; i < N; ++i) { mx += nb[i] / N; my += nb[N + i] / N; mt += nb[2 * N + i] / N; }
    if (fabs(mx - 10.0) > 0.1 || fabs(my) > 0.1 || fabs(mt) > 0.05 || !(bw[0] > 0 && bw[1] > 0 && bw[2] > 0)) {
      printf("FAIL clique up-solve: mean (%g, %g, %g), bw (%g, %g, %g)\n", mx, my, mt, bw[0], bw[1], bw[2]); return 1;
    }
    const int32_t upv_bad[1] = {0};   /* the rows target x1, the update list says x0 */
    u.up_var = upv_bad;
    if (rome_clique_upsolve(ctx, &ou, &u) != ROME_ERR_INVALID_ARG) { printf("FAIL up-solve argument check\n"); return 1; }
    u.up_var = upv;
    /* (6) the same up-solve device-resident: store <- beliefs once, a plan, a run; the new belief is written in place and mirrored
     * into a device buffer, from which a scatter plan fills a second store */
    rome_store *st = NULL, *st2 = NULL; rome_upsolve_plan* plan = NULL; rome_scatter_plan* sc = NULL;
    static double nb2[3 * N], back[3 * N], back2[3 * N];
    const int32_t mir[1] = {2};   /* block 2 of the mirror buffer */
    void* dmir = NULL;
    CHECK(rome_store_create(ctx, N, 2, 0, 0, &st));
    CHECK(rome_store_create(ctx, N, 2, 0, 0, &st2));
    CHECK(rome_store_upload(st, ROME_LAYOUT_SOA, 0, 0, 2, bel));
    CHECK(rome_dev_alloc(ctx, 3 * 6 * N * sizeof(double), &dmir));
    u.clique.bel_pose2 = NULL; u.new_pose2 = nb2; u.up_mirror = mir;
    CHECK(rome_upsolve_plan_create(ctx, st, &ou, &u, &plan));
    CHECK(rome_upsolve_plan_run(plan, &ou, (double*)dmir, 0));
    CHECK(rome_store_download(st, ROME_LAYOUT_SOA, 0, 1, 1, back));
    const int32_t sty[1] = {0}, sva[1] = {1}, sbl[1] = {2};
    CHECK(rome_scatter_plan_create(ctx, st2, 1, sty, sva, sbl, 0, &sc));
    CHECK(rome_scatter_plan_run(sc, (const double*)dmir));
    CHECK(rome_ctx_synchronize(ctx));
    CHECK(rome_store_download(st2, ROME_LAYOUT_SOA, 0, 1, 1, back2));
    for (int k = 0; k < 3 * N; ++k)
      if (nb2[k] != nb[k] || back[k] != nb[k] || back2[k] != nb[k]) { printf("FAIL plan != one-shot up-solve at %d: %g %g %g %g\n", k, nb[k], nb2[k], back[k], back2[k]); return 1; }
    if (rome_upsolve_plan_run(plan, &ou, NULL, 0) != ROME_ERR_INVALID_ARG) { printf("FAIL plan mirror check\n"); return 1; }
    rome_scatter_plan_destroy(sc); rome_upsolve_plan_destroy(plan); rome_store_destroy(st); rome_store_destroy(st2);
    CHECK(rome_dev_free(ctx, dmir));
  }
  /* (5) multihypo in the clique entry: pose x0, landmarks l1 = (10, 0), l2 = (30, 0) (test/testMultimodalRangeBearing.jl:27-53) */
  {
    static double bp[3 * N], bl[2 * 2 * N], outp[3 * N];
    rome_opts om; rome_opts_default(&om, ROME_SOLVER_NEWTON); om.n_particles = N; om.layout = ROME_LAYOUT_SOA; om.seed = 9;
    for (int i = 0; i < N; ++i) {
      bp[i] = 20.0 * sin(0.37 * i); bp[N + i] = sin(1.1 * i); bp[2 * N + i] = 0.05 * cos(0.3 * i);
      bl[i] = 10.0 + sin(2.0 * i); bl[N + i] = cos(1.7 * i); bl[2 * N + i] = 30.0 + cos(0.9 * i); bl[3 * N + i] = sin(0.8 * i);
    }
    const int32_t rows1[4] = {0, 1, 0, 0};   /* factor 0, dir 1 (solve the pose), fixed l1, target x0 */
    const int32_t alt1[1] = {1};              /* the other candidate: l2 */
    const double hw1[1] = {0.5}, mub[2] = {0.0, 20.0}, sgb[2] = {0.1, 1.0};
    rome_clique_host q; memset(&q, 0, sizeof(q));
    q.n_pose2 = 1; q.bel_pose2 = bp; q.n_point2 = 2; q.bel_point2 = bl;
    q.n_br1 = 1; q.f_br = 1; q.br1_rows4 = rows1; q.br_mu = mub; q.br_sigma = sgb; q.out_br1 = outp;
    q.br1_alt = alt1; q.br1_hypo_w = hw1;
    CHECK(rome_clique_proposals(ctx, &om, &q));
    int on1 = 0, on2 = 0;
    for (int i = 0; i < N; ++i) {
      const double d1 = hypot(outp[i] - 10.0, outp[N + i]), d2 = hypot(outp[i] - 30.0, outp[N + i]);
      on1 += fabs(d1 - 20.0) < 5.0; on2 += fabs(d2 - 20.0) < 5.0;
    }
    if (on1 < 20 || on1 > 80 || on2 < 20 || on2 > 80) { printf("FAIL multihypo clique row: %d on the l1 ring, %d on the l2 ring\n", on1, on2); return 1; }
    const int32_t alt_bad[1] = {2};
    q.br1_alt = alt_bad;
    if (rome_clique_proposals(ctx, &om, &q) != ROME_ERR_INVALID_ARG) { printf("FAIL multihypo argument check\n"); return 1; }
  }
  /* (7) messages between tree levels from plain C: block operations (anchor, relative) and a parent plan that takes (a) the child's
   *     anchor belief as a store-resident message and (b) a Pose2Pose2 row whose measurement samples are a block of the store.
   *     Store: block 0 = belief of the anchor variable a, 1 = its anchor copy, 2 = belief of s, 3 = relative samples, 4 = the parent's
   *     variable q, 5 = the child's message on a. */
  {
    static double b0[3 * N], b2[3 * N], back[3 * N], rel[3 * N];
    rome_store* st = NULL;
    CHECK(rome_store_create(ctx, N, 6, 0, 0, &st));
    for (int i = 0; i < N; ++i) {
      b0[i] = 2.0 + 0.1 * sin(0.9 * i); b0[N + i] = -1.0 + 0.1 * cos(1.3 * i); b0[2 * N + i] = 0.5 + 0.02 * sin(0.7 * i);
      b2[i] = 7.0 + 0.2 * cos(0.4 * i); b2[N + i] = 3.0 + 0.2 * sin(1.9 * i); b2[2 * N + i] = 1.0 + 0.03 * cos(0.6 * i);
    }
    CHECK(rome_store_upload(st, ROME_LAYOUT_SOA, 0, 0, 1, b0));
    CHECK(rome_store_upload(st, ROME_LAYOUT_SOA, 0, 2, 1, b2));
    CHECK(rome_store_upload(st, ROME_LAYOUT_SOA, 0, 5, 1, b0));
    const int32_t ty[1] = {0}, a0[1] = {0}, d1[1] = {1}, a1[1] = {1}, s2[1] = {2}, d3[1] = {3};
    rome_blockop_plan *anc = NULL, *rl = NULL;
    CHECK(rome_blockop_plan_create(ctx, st, ROME_BLOCKOP_ANCHOR, 1, ty, a0, NULL, d1, &anc));
    CHECK(rome_blockop_plan_create(ctx, st, ROME_BLOCKOP_RELATIVE, 1, ty, a1, s2, d3, &rl));
    CHECK(rome_blockop_plan_run(anc)); CHECK(rome_blockop_plan_run(rl)); CHECK(rome_ctx_synchronize(ctx));
    CHECK(rome_store_download(st, ROME_LAYOUT_SOA, 0, 1, 1, back));
    CHECK(rome_store_download(st, ROME_LAYOUT_SOA, 0, 3, 1, rel));
    double mx = 0, my = 0, ss = 0, cc = 0;
    for (int i = 0; i < N; ++i) { mx += b0[i]; my += b0[N + i]; ss += sin(b0[2 * N + i]); cc += cos(b0[2 * N + i]); }
    mx /= N; my /= N; const double mt = atan2(ss, cc);
    for (int i = 0; i < N; ++i) {
      if (fabs(back[i] - mx) > 1e-12 || fabs(back[N + i] - my) > 1e-12 || fabs(back[2 * N + i] - mt) > 1e-12) { printf("FAIL anchor block at %d\n", i); return 1; }
      const double dx = b2[i] - mx, dy = b2[N + i] - my;   /* anchor (+) rel_i must give particle i of s back */
      const double ex = cos(mt) * dx + sin(mt) * dy, ey = -sin(mt) * dx + cos(mt) * dy;
      if (fabs(rel[i] - ex) > 1e-12 || fabs(rel[N + i] - ey) > 1e-12) { printf("FAIL relative samples at %d\n", i); return 1; }
    }
    /* parent: q = a (+) z with z the relative samples (sampled-measurement row), beside the child's message on a (store-resident message) */
    rome_opts op; rome_opts_default(&op, ROME_SOLVER_NEWTON); op.n_particles = N; op.layout = ROME_LAYOUT_SOA; op.seed = 3;
    const int32_t rows[2 * 4] = {0, 0, 0, 4,   0, 1, 4, 0};   /* a -> q (dir 0), then q -> a (dir 1): both rows read block 3 as their samples */
    const int32_t meas[2] = {3, 3};
    const double mu0[3] = {0, 0, 0}, cov0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const int32_t upt[2] = {0, 0}, upv[2] = {4, 0}, msrc[1] = {5}, mup[1] = {1};
    rome_clique_upsolve_host u; memset(&u, 0, sizeof(u));
    u.clique.n_p2p2 = 2; u.clique.f_p2p2 = 1; u.clique.p2p2_rows4 = rows; u.clique.p2p2_mu = mu0; u.clique.p2p2_cov = cov0; u.clique.p2p2_meas = meas;
    u.gibbs_iters = 1; u.product_iters = 1; u.schedule = ROME_UPSOLVE_SEQUENTIAL; u.n_up = 2; u.up_type = upt; u.up_var = upv;
    u.n_smsg_pose2 = 1; u.smsg_pose2_src = msrc; u.smsg_pose2_up = mup;
    rome_upsolve_plan* pl = NULL;
    CHECK(rome_upsolve_plan_create(ctx, st, &op, &u, &pl));
    CHECK(rome_upsolve_plan_run(pl, &op, NULL, 0)); CHECK(rome_ctx_synchronize(ctx));
    CHECK(rome_store_download(st, ROME_LAYOUT_SOA, 0, 4, 1, back));
    for (int i = 0; i < N; ++i) {   /* one proposal (K = 1): q_i = a_i (+) rel_i exactly */
      const double ax = b0[i], ay = b0[N + i], at = b0[2 * N + i];
      const double qx = ax + cos(at) * rel[i] - sin(at) * rel[N + i], qy = ay + sin(at) * rel[i] + cos(at) * rel[N + i];
      if (fabs(back[i] - qx) > 1e-9 || fabs(back[N + i] - qy) > 1e-9) { printf("FAIL sampled-measurement row at %d: %g %g vs %g %g\n", i, back[i], back[N + i], qx, qy); return 1; }
    }
    CHECK(rome_store_download(st, ROME_LAYOUT_SOA, 0, 0, 1, back));   /* a: product of the q -> a proposal and the stored message: stays near a */
    double ma = 0; for (int i = 0; i < N; ++i) ma += back[i]; ma /= N;
    if (!(fabs(ma - mx) < 0.5)) { printf("FAIL product with a store-resident message: mean %g vs %g\n", ma, mx); return 1; }
    const int32_t bad[1] = {0};   /* a message source must not be a variable the plan updates */
    u.smsg_pose2_src = bad;
    rome_upsolve_plan* pb = NULL;
    if (rome_upsolve_plan_create(ctx, st, &op, &u, &pb) != ROME_ERR_INVALID_ARG) { printf("FAIL store-message argument check\n"); return 1; }
    rome_upsolve_plan_destroy(pl); rome_blockop_plan_destroy(anc); rome_blockop_plan_destroy(rl); rome_store_destroy(st);
  }
  /* (8) stream order through ONE context: milliseconds of block operations queued on the context's private (non-blocking) stream, then
   *     rome_ctx_set_stream(NULL) and a block operation + download on HIP's null stream -- which nothing joins to a non-blocking stream
   *     unless rome_ctx_set_stream orders the stream change itself.  The copy on the new stream must see what the old stream wrote
   *     (round 5's race: TreeSolver's level-0 anchor operation on the private stream against the first sharded level plan on the
   *     caller's stream; no context workspace is involved, so the old "drain if a workspace is in use" rule did not fire).
   *     The reference run drains the context between every two calls.  Against the round-5 library this check FAILS
   *     (profiles/r06_stream_order.txt). */
  {
    enum { M = 3000, REP = 400 };
    static double b0[3 * N], seq[3 * N], ord[3 * N];
    static int32_t ty[M], src[M], dst[M];
    for (int i = 0; i < N; ++i) { b0[i] = 1.0 + 0.1 * sin(0.9 * i); b0[N + i] = -2.0 + 0.1 * cos(1.3 * i); b0[2 * N + i] = 0.3 + 0.02 * sin(0.7 * i); }
    for (int k = 0; k < M; ++k) { ty[k] = 0; src[k] = k; dst[k] = k + 1; }   /* block k+1 <- N copies of the mean of block k: a chain inside one launch is a race, so ... */
    for (int k = 0; k < M; ++k) src[k] = 0;                                   /* ... every block takes the mean of block 0 */
    const int32_t t1[1] = {0}, last[1] = {M}, dA[1] = {M + 1}, sA[1] = {M + 1}, dB[1] = {M + 2};
    for (int pass = 0; pass < 2; ++pass) {   /* 0: drained between the calls (the reference), 1: back to back across the stream change */
      rome_store* st = NULL; rome_blockop_plan *big = NULL, *anc = NULL, *cp = NULL;
      CHECK(rome_ctx_use_own_stream(ctx));
      CHECK(rome_store_create(ctx, N, M + 3, 0, 0, &st));
      CHECK(rome_store_upload(st, ROME_LAYOUT_SOA, 0, 0, 1, b0));
      CHECK(rome_blockop_plan_create(ctx, st, ROME_BLOCKOP_ANCHOR, M, ty, src, NULL, dst, &big));   /* blocks 1..M <- mean of block 0 */
      CHECK(rome_blockop_plan_create(ctx, st, ROME_BLOCKOP_ANCHOR, 1, t1, last, NULL, dA, &anc));   /* mean of block M -> block M+1 */
      CHECK(rome_blockop_plan_create(ctx, st, ROME_BLOCKOP_COPY, 1, t1, sA, NULL, dB, &cp));        /* block M+1 -> block M+2 */
      CHECK(rome_ctx_synchronize(ctx));
      for (int rep = 0; rep < REP; ++rep) CHECK(rome_blockop_plan_run(big));                          /* private stream: the queue fills */
      if (pass == 0) CHECK(rome_ctx_synchronize(ctx));
      CHECK(rome_blockop_plan_run(anc));                                                              /* private stream, behind the queue */
      if (pass == 0) CHECK(rome_ctx_synchronize(ctx));
      CHECK(rome_ctx_set_stream(ctx, NULL));
      CHECK(rome_blockop_plan_run(cp));                                                               /* null stream */
      CHECK(rome_store_download(st, ROME_LAYOUT_SOA, 0, M + 2, 1, pass == 0 ? seq : ord));           /* null stream */
      CHECK(rome_ctx_use_own_stream(ctx));
      CHECK(rome_ctx_synchronize(ctx));
      rome_blockop_plan_destroy(big); rome_blockop_plan_destroy(anc); rome_blockop_plan_destroy(cp); rome_store_destroy(st);
    }
    if (!(fabs(seq[0] - 1.0) < 0.1 && fabs(seq[N] + 2.0) < 0.1)) { printf("FAIL stream-order reference: anchor mean %g %g\n", seq[0], seq[N]); return 1; }
    for (int k = 0; k < 3 * N; ++k)
      if (seq[k] != ord[k]) { printf("FAIL rome_ctx_set_stream did not order the new stream after the old one: [%d] %.17g vs %.17g\n", k, seq[k], ord[k]); return 1; }
  }
  rome_ctx_destroy(ctx);
  printf("abi_smoke ok (max |Δ| vs closed form %.2e)\n", worst);
  return 0;
}
