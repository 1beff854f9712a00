/* rome_mi355.h -- C ABI of librome_mi355.so: MI355X (gfx950) factor convolutions for RoME.jl.
 *
 * Drop-in boundary for ONE hot path of RoME.jl + IncrementalInference.jl: the per-particle
 * residual + numerical root-find inside `approxConvBelief` (IIF `computeAcrossHypothesis!` ->
 * `_solveCCWNumeric!` -> `_solveLambdaNumeric`) for the factors
 *     Pose2Pose2                 src/factors/Pose2D.jl:30-67
 *     PriorPose2                 src/factors/PriorPose2.jl:13-47
 *     Pose2Point2BearingRange    src/factors/BearingRange2D.jl:10-64
 *     Pose3Pose3                 src/factors/Pose3Pose3.jl:9-29      (+ PriorPose3, src/factors/Pose3D.jl:15-19)
 * (paths relative to the RoME.jl v0.24.6 checkout).  Each entry point names the reference
 * interface it replaces.  Plain pointers and sizes only; no exceptions cross the ABI; the library
 * never retains caller pointers past a call.  There is NO CPU fallback: without a HIP device
 * rome_ctx_create fails with ROME_ERR_NO_DEVICE.
 *
 * Conventions
 *   - FP64 everywhere (the reference computes in Float64).
 *   - Pose2 coordinates (x, y, θ)  <->  point ((x,y), R(θ))      (src/variables/VariableTypes.jl:35)
 *     Point2 coordinates (x, y)                                  (src/variables/VariableTypes.jl:13)
 *     Pose3 coordinates (x, y, z, ωx, ωy, ωz) <-> (t, Exp(ω))    (src/variables/VariableTypes.jl:47,
 *                                                                 coordinate order src/services/g2oParser.jl:166)
 *   - a "block" is one belief of N particles; ROME_LAYOUT_SOA: [block][dim][N] (device-native),
 *     ROME_LAYOUT_AOS: [block][N][dim] (what a Julia Vector of coordinate SVectors looks like);
 *     ROME_LAYOUT_AOS_POINTS: [block][N][point_len], the reference's NATIVE point containers (Pose2
 *     ArrayPartition = [tx,ty,R11,R21,R12,R22], Point2 = [x,y], Pose3 = [t(3), R column-major(9)]): a Julia
 *     caller passes pointer(getVal(...)) unchanged; noise blocks stay [block][N][dz] coordinates.
 *   - dir = 0 solves for the factor's 2nd variable given the 1st (x_i -> x_j / pose -> landmark),
 *     dir = 1 solves for the 1st given the 2nd.  In the Pose2Pose2 / Pose3Pose3 tables dir = 2
 *     (ROME_DIR_PRIOR) marks a PriorPose2 / PriorPose3 row: no fixed variable, the proposal is the
 *     prior sample itself, so a whole-graph sweep (relative factors + priors) is ONE launch.
 *   - return value: ROME_OK (0) or a negative ROME_ERR_*; per-particle non-convergence is reported
 *     through the optional `status` array (0 = converged, 1 = iteration cap reached).
 *   - a rome_ctx is not thread-safe; distinct contexts are (one HIP stream each).
 */
#ifndef ROME_MI355_H
#define ROME_MI355_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROME_MI355_VERSION 122 /* 0.1.22: ROME_BLOCKOP_COMPOSE / MIX, rome_blockop_plan_create_ex (round 6: solveTree as variable elimination in relative-factor algebra); rome_ctx_set_stream orders streams by event.  0.1.21: store-resident messages (smsg_*) in rome_clique_upsolve_host (a tree level reads its children's separator beliefs from HBM); multihypo / nullhypo / stream-id columns in rome_clique_host; rome_store + rome_upsolve_plan
                                 * (device-resident clique up-solves: beliefs stay in HBM across frontiers) */

enum {
  ROME_OK = 0,
  ROME_ERR_INVALID_ARG = -1,
  ROME_ERR_NO_DEVICE = -2,
  ROME_ERR_HIP = -3,         /* see rome_last_hip_error() */
  ROME_ERR_NOT_POSDEF = -4,  /* a covariance has no Cholesky factor */
  ROME_ERR_UNSUPPORTED_N = -5, /* n_particles > ROME_MAX_PARTICLES */
  ROME_ERR_ALLOC = -6
};

#define ROME_MAX_PARTICLES 4096   /* convolutions / prior sampling: N <= 512 lives in registers, larger N is walked in chunks of 128;
                                    the KDE and importance-product entries take N <= 512, the Gibbs product N <= 256 */
#define ROME_MAX_PARTICLES_REGISTER 512
/* per-stage particle limits of ONE solve iteration (every entry fails with ROME_ERR_UNSUPPORTED_N / ROME_ERR_INVALID_ARG above its
 * limit, nothing is truncated; rome_jl_amd.solveGraph / DeviceGraph.solve check them before the first launch):
 *   convolutions / prior sampling  ROME_MAX_PARTICLES (4096)      manikde! bandwidths, getKDEMax   ROME_MAX_PARTICLES_KDE (512)
 *   importance product             ROME_MAX_PARTICLES_PRODUCT (512; Pose3: 256)
 *   multiscale Gibbs product, rome_clique_upsolve, up-solve plans ROME_MAX_PARTICLES_GIBBS (256) */
#define ROME_MAX_PARTICLES_KDE 512
#define ROME_MAX_PARTICLES_PRODUCT 512
#define ROME_MAX_PARTICLES_PRODUCT_POSE3 256
#define ROME_MAX_PARTICLES_GIBBS 256   /* two kernel instantiations: N <= 128 (128-thread blocks, what the solve-loop numbers are quoted on) and N <= 256 */

/* Solvers for the per-particle root-find (replaces Optim.optimize(cost, X0c, NelderMead()) in IIF
 * `_solveLambdaNumeric`, called for every particle of every convolution):
 *   CLOSED_FORM   analytic root, status always 0; start point / inflation only matter for the
 *                 under-determined bearing-range -> pose direction (a ring of roots: the member nearest the jittered start)
 *   NEWTON        (default) unique-root factors (Pose2Pose2, PriorPose2, bearing-range -> landmark, Pose3Pose3): the root of the
 *                 residual is unique, so no start point or inflation cycle can change it: the kernel returns the analytic root
 *                 and, when a `status` array is given, evaluates the residual FUNCTOR there (status = max|r| <= tol ? 0 : 1).
 *                 The start points are not read.  Bearing-range -> pose (a ring of roots): inflate_cycles x {entropy, the exact
 *                 step onto the ring member the jittered start selects} -- the same points as CLOSED_FORM -- and the functor at
 *                 the final point for `status`.
 *   NELDER_MEAD   Optim.jl's NelderMead() with its defaults on Σ r², i.e. the reference's algorithm, inflate_cycles x
 *                 {entropy, minimise} from the start points
 *   GAUSS_NEWTON  numerical root-find on the residual of the reference's CalcFactor, evaluated at every iterate, with analytic group
 *                 updates, iterated to max|r| <= tol.  The rotation part of an iterate is carried in the form the residual needs --
 *                 (cos, sin) of a Pose2 heading, a unit quaternion of a Pose3 rotation: r_w = Log(conj(q_q) (x) q_p (x) q_z) is the
 *                 rotation of the functor's Log(R_q^T R_p Exp(z_w)) -- and its coordinates (atan2 / Log) are evaluated where the
 *                 step or the convergence test needs them (NEWTON's `status` evaluates the same residual once, at the returned
 *                 root; the literal point / 3x3 forms are the rome_residual_* entry points).  Unique-root
 *                 factors: from the start points (the target's current belief), ONE pass -- a converged unique root does not
 *                 depend on the start beyond `tol`, so the entropy / re-solve rounds of inflate_cycles are not run (a converged
 *                 proposal is start- and cycle-count-independent only to `tol`: compared with the oracle's Newton mode, which
 *                 runs every cycle, to 1e-9).  Bearing-range -> pose: inflate_cycles x {entropy, iterate}, as the oracle.      */
enum { ROME_SOLVER_CLOSED_FORM = 0, ROME_SOLVER_NEWTON = 1, ROME_SOLVER_NELDER_MEAD = 2, ROME_SOLVER_GAUSS_NEWTON = 3 };
enum { ROME_LAYOUT_SOA = 0, ROME_LAYOUT_AOS = 1, ROME_LAYOUT_AOS_POINTS = 2 };
enum { ROME_DIR_TO = 0, ROME_DIR_FROM = 1, ROME_DIR_PRIOR = 2 };
enum { ROME_NOISE_STANDARD_NORMALS = 0, ROME_NOISE_MEASUREMENTS = 1 };

/* Mirrors the IIF SolverParams fields that reach this path (N, inflateCycles, inflation). */
typedef struct rome_opts {
  int32_t n_particles;    /* N, IIF default 100 (src/canonical/GenerateHexagonal.jl:30)               */
  int32_t solver;         /* ROME_SOLVER_*                                                             */
  int32_t max_iters;      /* NEWTON / GAUSS_NEWTON default 20 ; NELDER_MEAD default 1000 (Optim iterations) */
  int32_t inflate_cycles; /* IIF inflateCycles, default 3                                              */
  double  tol;            /* NEWTON / GAUSS_NEWTON: max|r| <= tol (1e-12) ; NELDER_MEAD: Optim g_tol (1e-8) */
  double  inflation;      /* IIF inflation (kappa) for the entropy added before each cycle, default 5.0 */
  uint64_t seed;          /* Philox4x32-10 key                                                         */
  uint64_t stream_offset; /* Philox stream of convolution c = stream_offset + c  (global conv id when sharded) */
  int32_t layout;         /* host-pointer entry points only: ROME_LAYOUT_*                             */
  int32_t presampled;     /* meaning of the `noise` argument / field when it is given:
                           *   ROME_NOISE_STANDARD_NORMALS (0): ξ ~ N(0, I), the library forms z = μ + Lξ (bearing-range: μ + σ ξ);
                           *   ROME_NOISE_MEASUREMENTS (1): the rows ARE the measurement samples z (tangent coordinates of the
                           *   factor's `getSample`): any `SamplableBelief` -- Rayleigh, mixtures, AliasingScalarSampler, a KDE --
                           *   sampled by the caller (`Pose2Point2BearingRange{B,R}` is generic in B and R,
                           *   src/factors/BearingRange2D.jl:10-27); mu / cov / sigma are then unused                       */
  double  spread_nh;      /* IIF spreadNH (default 3.0): entropy scale for particles of the other hypothesis (multihypo) */
  double  nullhypo;       /* host-pointer entry points only: IIF nullhypo= probability applied to every row of the call */
} rome_opts;

typedef struct rome_ctx rome_ctx; /* opaque: device id, HIP stream, staging buffers */

int  rome_version(void);
const char* rome_strerror(int code);
int  rome_last_hip_error(const rome_ctx* ctx); /* raw hipError_t of the last failing HIP call */
const char* rome_last_hip_error_string(const rome_ctx* ctx);

void rome_opts_default(rome_opts* o, int32_t solver);
/* (SURVEY §8(b) sketched `device = -1 -> CPU oracle`: deliberately NOT provided -- the library has no CPU path of any kind; without a
 * HIP device rome_ctx_create returns ROME_ERR_NO_DEVICE) */
int  rome_ctx_create(rome_ctx** out, int device);
void rome_ctx_destroy(rome_ctx* ctx);
int  rome_ctx_set_stream(rome_ctx* ctx, void* hip_stream); /* launch on a caller-owned hipStream_t; NULL = HIP's default (null) stream.
                                                             * When the stream CHANGES, the new stream is ordered after everything queued
                                                             * on the previous one (event record + stream wait, no host synchronisation):
                                                             * what is launched through ONE context is one sequence whatever the stream.
                                                             * Concurrency takes one context per stream (SeparatorPipeline: one per slot) */
int  rome_ctx_use_own_stream(rome_ctx* ctx);               /* back to the context's private non-blocking stream (the default) */
int  rome_ctx_synchronize(rome_ctx* ctx);
int  rome_device_count(void);

/* Native point containers <-> coordinates, n rows (dim 3: Pose2 6 doubles, dim 6: Pose3 12 doubles, dim 2: copy):
 * getCoordinates(Pose2, p) = vee(log(ϵ, p)) and getPoint(Pose2, c) = exp_ϵ(hat(c)) of the reference
 * (src/variables/VariableTypes.jl:35,47), evaluated on the device.  Host pointers.                    */
int rome_points_to_coords(rome_ctx*, int32_t dim, int32_t n, const double* pts, double* coords);
int rome_coords_to_points(rome_ctx*, int32_t dim, int32_t n, const double* coords, double* pts);

/* Σ (d x d row-major, n of them) -> row-packed lower Cholesky factors (n x d(d+1)/2), host side.
 * Replaces the PDMat factorisation MvNormal(μ, Σ) performs at factor construction
 * (src/services/g2oParser.jl:103-121).                                                           */
int rome_cholesky_lower(int32_t d, int32_t n, const double* cov, double* L);

/* ---------------------------------------------------------------------------------------------
 * Residual-only entry points (host pointers, n rows of coordinates).  Replace one call of the
 * CalcFactor functor each:
 *   rome_residual_pose2pose2      (cf::CalcFactor{<:Pose2Pose2})(X, p, q)            Pose2D.jl:51-67
 *   rome_residual_priorpose2      (cf::CalcFactor{<:PriorPose2})(m, p)               PriorPose2.jl:37-47
 *   rome_residual_pose2point2br   (cf::CalcFactor{<:Pose2Point2BearingRange})(z,p,l) BearingRange2D.jl:48-64
 *   rome_residual_pose3pose3      (cf::CalcFactor{<:Pose3Pose3})(X, p, q)            Pose3Pose3.jl:17-29
 *   rome_residual_priorpose3      (cf::CalcFactor{<:PriorPose3})(m, p)               Pose3D.jl:15-19
 * z: measurement tangent coordinates; p, q, m: pose coordinates; l: landmark.
 * The *_pt variants take the reference's native point layouts (Pose2: [tx,ty,R11,R21,R12,R22];
 * Pose3: [t(3), R column-major(9)]) so a Julia caller can pass pointer(vals) without conversion. */
int rome_residual_pose2pose2(rome_ctx*, int32_t n, const double* z /*n*3*/, const double* p /*n*3*/, const double* q /*n*3*/, double* r /*n*3*/);
int rome_residual_priorpose2(rome_ctx*, int32_t n, const double* m /*n*3*/, const double* p /*n*3*/, double* r /*n*3*/);
int rome_residual_pose2point2br(rome_ctx*, int32_t n, const double* z /*n*2 (bearing,range)*/, const double* p /*n*3*/, const double* l /*n*2*/, double* r /*n*2*/);
int rome_residual_pose2point2br_pt(rome_ctx*, int32_t n, const double* z /*n*2*/, const double* p_pt /*n*6*/, const double* l /*n*2*/, double* r /*n*2*/);
int rome_residual_pose3pose3(rome_ctx*, int32_t n, const double* z /*n*6*/, const double* p /*n*6*/, const double* q /*n*6*/, double* r /*n*6*/);
int rome_residual_pose3pose3_pt(rome_ctx*, int32_t n, const double* z /*n*6*/, const double* p_pt /*n*12*/, const double* q_pt /*n*12*/, double* r /*n*6*/);
int rome_residual_priorpose3(rome_ctx*, int32_t n, const double* m /*n*6*/, const double* p /*n*6*/, double* r /*n*6*/);

/* ---------------------------------------------------------------------------------------------
 * Batched factor convolutions, HOST pointers: C independent convolutions x N particles.
 * Each replaces C calls of IIF `approxConvBelief(dfg, factor, target)` up to (not including) the
 * `manikde!` wrap: N x getSample, then inflate_cycles x { addEntropyOnManifold!, N x _solveCCWNumeric! }.
 *   dir    [C] or NULL (all 0)
 *   mu     [C][dz]   measurement mean             cov [C][dz*dz] row-major covariance Σ
 *   fixed  C blocks of the fixed variable's particles (layout per opts->layout)
 *   noise  C blocks [.][dz] of standard-normal ξ (z = μ + chol(Σ) ξ), or NULL -> in-kernel Philox
 *   target_inout  C blocks: start points u0 (current belief of the target) in, solutions out
 *   status [C][N] or NULL                                                                        */
int rome_conv_pose2pose2(rome_ctx*, const rome_opts*, int32_t C, const int32_t* dir,
                         const double* mu /*C*3*/, const double* cov /*C*9*/,
                         const double* fixed /*C*N*3*/, const double* noise /*C*N*3 or NULL*/,
                         double* target_inout /*C*N*3*/, int32_t* status);
/* dir (scalar): 0 = fixed poses (3) -> target landmarks (2); 1 = fixed landmarks (2) -> target poses (3).
 * mu = (mean bearing, mean range), sigma = (σ_b, σ_ρ) of the two Normal() fields
 * (src/factors/BearingRange2D.jl:10-13); getSample :17-27.  A negative sigma encodes a Uniform belief on
 * [mu - |sigma|, mu + |sigma|] (test/TestPoseAndPoint2Constraints.jl:95 uses Uniform(-pi,pi) bearings).                                      */
/* Pose2Pose2 with IIF `multihypo=[1, w, 1-w]` over the SECOND pose of the factor (addFactor!(fg, [:a; :b1; :b2], Pose2Pose2(z),
 * multihypo=[1; w; 1-w]): IIF accepts the keyword on any factor; the reference's own uses are on bearing-range factors,
 * test/testMultimodalRangeBearing.jl:53 -- the same rule is applied here).  One direction per call.  dir 1 (solve a): `fixed` holds
 * the particles of b1, `alt` those of b2, per particle the fixed pose is drawn from (b1 with probability hypo_w[c], else b2).
 * dir 0 (solve b1 from a): `alt` = b2; particles drawn for b2 are not constrained by the factor, they keep their value and receive
 * spreadNH * |mean_xy(b1) - mean_xy(b2)| * (U - 1/2) entropy on every coordinate. */
int rome_conv_pose2pose2_mh(rome_ctx*, const rome_opts*, int32_t C, int32_t dir,
                            const double* mu /*C*3*/, const double* cov /*C*9*/,
                            const double* fixed /*C*N*3*/, const double* alt /*C*N*3*/, const double* hypo_w /*C*/,
                            const double* noise /*C*N*3 or NULL*/, double* target_inout /*C*N*3*/, int32_t* status);
int rome_conv_pose2point2br(rome_ctx*, const rome_opts*, int32_t C, int32_t dir,
                            const double* mu /*C*2*/, const double* sigma /*C*2*/,
                            const double* fixed, const double* noise /*C*N*2 or NULL*/,
                            double* target_inout, int32_t* status);
/* same with IIF `multihypo=[1, w, 1-w]` over two landmark candidates (addFactor!(fg, [:x0;:l1;:l2], p2br,
 * multihypo=[1.0;0.5;0.5]), test/testMultimodalRangeBearing.jl:53): `alt` holds C blocks of the OTHER landmark's
 * particles, hypo_w[c] the probability of the primary one.  dir 1: per particle the fixed landmark is drawn from
 * (primary, alt); dir 0: target_inout is the primary landmark, particles drawn for `alt` only get spreadNH entropy. */
int rome_conv_pose2point2br_mh(rome_ctx*, const rome_opts*, int32_t C, int32_t dir,
                               const double* mu /*C*2*/, const double* sigma /*C*2*/,
                               const double* fixed, const double* alt /*C*N*2*/, const double* hypo_w /*C*/,
                               const double* noise /*C*N*2 or NULL*/, double* target_inout, int32_t* status);
int rome_conv_pose3pose3(rome_ctx*, const rome_opts*, int32_t C, const int32_t* dir,
                         const double* mu /*C*6*/, const double* cov /*C*36*/,
                         const double* fixed /*C*N*6*/, const double* noise /*C*N*6 or NULL*/,
                         double* target_inout /*C*N*6*/, int32_t* status);
/* Prior "convolution" = N samples of the prior as points: IIF samplePoint on PriorPose2.Z / PriorPose3.Z
 * (src/factors/PriorPose2.jl:13-17, src/factors/Pose3D.jl:8-12).                                 */
int rome_sample_priorpose2(rome_ctx*, const rome_opts*, int32_t C, const double* mu /*C*3*/, const double* cov /*C*9*/,
                           const double* noise /*C*N*3 or NULL*/, double* out /*C*N*3*/);
int rome_sample_priorpose3(rome_ctx*, const rome_opts*, int32_t C, const double* mu /*C*6*/, const double* cov /*C*36*/,
                           const double* noise /*C*N*6 or NULL*/, double* out /*C*N*6*/);
/* PriorPoint2 (src/factors/Point2D.jl:8-18) in the NON-parametric path: N samples of the landmark prior MvNormal(mu, cov) -- the
 * proposal a landmark prior contributes to the product of its variable (test/testBearingRange2D.jl:324 puts one on the landmark of the
 * "solve for pose" test). */
int rome_sample_priorpoint2(rome_ctx*, const rome_opts*, int32_t C, const double* mu /*C*2*/, const double* cov /*C*4*/,
                            const double* noise /*C*N*2 or NULL*/, double* out /*C*N*2*/);

/* ---------------------------------------------------------------------------------------------
 * Clique-level batch from HOST beliefs: every (factor, direction) convolution of a clique -- or of one variable, what IIF
 * `proposalbeliefs!(dfg, destlbl, factors, ...)` computes factor by factor (each an `approxConvBelief`; the call
 * behind `solveTree!`, examples/ManhattanDatasetBatch.jl:43) -- in ONE call: the beliefs are passed once per VARIABLE, the row
 * tables index them, one kernel launch per factor family, one synchronisation.  A per-factor call moves 3 belief blocks and pays
 * ~45 µs of latency per convolution; this entry moves every belief once (PCIe-inclusive rate: DESIGN.md §6).
 *   bel_*      [n][dim][N] blocks per variable type (layout per opts->layout: SoA, AoS [n][N][dim], or the reference's point
 *              containers for Pose2 / Pose3), n_* variables of the type in the clique
 *   *_rows4    [rows][4] = (factor, dir, fixed_var, target_var): indices into the family's factor table and the belief arrays
 *              (Pose2Pose2 / Pose3Pose3: dir 0 solves the 2nd variable, 1 the 1st, ROME_DIR_PRIOR = prior row with
 *              fixed_var = target_var; bearing-range: br1 rows solve the pose from the landmark, br0 the landmark from the pose)
 *   *_mu/_cov  factor tables ([F][dz], [F][dz*dz] row-major covariances; bearing-range: [F][2] sigmas, < 0 = Uniform half-width)
 *   out_*      [rows][dt][N] proposals in the same layout as the beliefs
 * Philox stream of row r of a family = opts->stream_offset + family offset (0, 1<<28, 2<<28, 5<<28, 7<<28 for p2p2, br1, br0, p3p3, prpt2: the
 * same as the device-resident graph sweeps) + r, so a clique call reproduces the per-factor calls made with those streams.  */
typedef struct rome_clique_host {
  int32_t n_pose2, n_point2, n_pose3, reserved0;
  const double* bel_pose2; const double* bel_point2; const double* bel_pose3;
  int32_t n_p2p2, f_p2p2; const int32_t* p2p2_rows4; const double* p2p2_mu; const double* p2p2_cov; double* out_p2p2;
  int32_t n_br1, n_br0; int32_t f_br, reserved1; const int32_t* br1_rows4; const int32_t* br0_rows4;
  const double* br_mu; const double* br_sigma; double* out_br1; double* out_br0;
  int32_t n_p3p3, f_p3p3; const int32_t* p3p3_rows4; const double* p3p3_mu; const double* p3p3_cov; double* out_p3p3;
  /* landmark priors (PriorPoint2, src/factors/Point2D.jl:8-18): rows (factor, ROME_DIR_PRIOR, var, var) over bel_point2; mu [F][2],
   * cov [F][4]; proposals [rows][2][N]; Philox family offset 7 << 28.  In rome_clique_upsolve their proposals follow the
   * bearing-range -> landmark rows in the product of their landmark. */
  int32_t n_prpt2, f_prpt2; const int32_t* prpt2_rows4; const double* prpt2_mu; const double* prpt2_cov; double* out_prpt2;
  /* optional per-row hypothesis columns (NULL = none), as IIF attaches them per factor: `multihypo=[1, w, 1-w]`
   * (addFactor!(fg, [:x0;:l1;:l2], p2br, multihypo=[1.0;0.5;0.5]), test/testMultimodalRangeBearing.jl:53,
   * examples/MultimodalRangeBearing.jl:33) and `nullhypo=p` (test/testPose3Pose3NH.jl:118); IIF resolves both inside the same
   * proposalbeliefs! / upGibbsCliqueDensity loop (computeAcrossHypothesis!).
   *   <fam>_alt [rows]      index of the OTHER candidate of the factor's second variable, -1 = ordinary row: for rows that solve the
   *                         first variable (p2p2 dir 1, br1) it indexes the belief array of the FIXED side (the fixed particle is drawn
   *                         per particle from (fixed_var with probability hypo_w, alt)); for rows that solve a candidate (p2p2 dir 0,
   *                         br0) the array of the TARGET side (particles drawn for the other candidate keep their value + spreadNH entropy)
   *   <fam>_hypo_w [rows]   probability that the row's own candidate is the one the measurement belongs to (required with _alt)
   *   <fam>_nullhypo [rows] probability that the factor does not apply to a particle (0 = ordinary)                              */
  const int32_t* p2p2_alt; const double* p2p2_hypo_w; const double* p2p2_nullhypo;
  const int32_t* br1_alt;  const double* br1_hypo_w;  const double* br1_nullhypo;
  const int32_t* br0_alt;  const double* br0_hypo_w;  const double* br0_nullhypo;
  const double* p3p3_nullhypo;
  /* optional per-row Philox stream ids (NULL = the row's index in its table): row r of a family draws stream
   * opts->stream_offset + family offset + <fam>_stream[r] (0 <= id < 2^28).  A clique / frontier that is a SUBSET of a larger table
   * (one rank's share of a frontier, a clique of a graph-wide table) then draws exactly what the whole table draws: results do not
   * depend on how the work was partitioned. */
  const int32_t* p2p2_stream; const int32_t* br1_stream; const int32_t* br0_stream; const int32_t* p3p3_stream; const int32_t* prpt2_stream;
  /* optional [n_p2p2], rome_upsolve_plan only: p2p2_meas[r] >= 0 makes row r a Pose2Pose2 factor whose measurement distribution is a set
   * of N SAMPLES (IIF accepts any SamplableBelief as `Z`): the N tangent coordinates (x, y, theta) of Pose2 block p2p2_meas[r] of the
   * plan's store, particle i of the fixed variable taking sample i; the row's factor entry is ignored.  -1: an ordinary row.  This is
   * how the RELATIVE up-message of a child clique (samples of anchor^-1 * separator, rome_blockop RELATIVE) enters its parent's solve. */
  const int32_t* p2p2_meas;
  /* the same for bearing-range rows: br1_meas / br0_meas [rows] = POINT2 block whose N (x, y) entries are the row's (bearing, range)
   * samples -- the relative message pose -> landmark of a child clique (ROME_BLOCKOP_RELATIVE with a Point2 source) */
  const int32_t* br1_meas; const int32_t* br0_meas;
} rome_clique_host;
int rome_clique_proposals(rome_ctx*, const rome_opts*, const rome_clique_host*);

/* ---------------------------------------------------------------------------------------------
 * Clique up-solve, device-resident: IIF `upGibbsCliqueDensity` (the loop behind every clique of `solveTree!`,
 * examples/ManhattanDatasetBatch.jl:43, src/services/AdditionalUtils.jl:18-19):
 *     gibbsIters x  for each variable to update (the clique's frontals, in order):
 *                       proposalbeliefs! (one approxConvBelief per factor of the variable)  ->  manikde! of every proposal
 *                       -> manifoldProduct -> setValKDE!
 * in ONE call: beliefs and tables cross PCIe once, then gibbs_iters x {one convolution launch per factor family, manikde!
 * bandwidths of the proposals, multiscale Gibbs product, write-back into the device belief store} run back to back on the
 * context's stream; the new beliefs of the updated variables and their manikde! bandwidths come back at the end.
 *   clique      beliefs of ALL variables of the clique (frontals + separators) and the family row tables, exactly as for
 *               rome_clique_proposals (out_* are ignored and may be NULL).  Every row must TARGET an updated variable, and the rows
 *               of each family must be grouped by target in the order of the update list (checked: ROME_ERR_INVALID_ARG).
 *   up_type/up_var [n_up]   the variables to update, in Gibbs order: type 0 Pose2 / 1 Point2 / 2 Pose3, index into bel_<type>
 *   schedule    ROME_UPSOLVE_SEQUENTIAL: IIF's order -- one variable at a time, each seeing the beliefs already updated in this
 *               iteration; ROME_UPSOLVE_JACOBI: all updated variables from the beliefs of the previous iteration (fewer launches)
 *   msg_<type>  optional upward messages of child cliques: n_msg_<type> extra densities of N points each ([.][dim][N], layout per
 *               opts) on the updated variable at position msg_<type>_up[m] of the update list; they enter every product of
 *               that variable beside the factor proposals (their manikde! bandwidths are computed once)
 *   new_<type>  [number of updated variables of the type, in update order][dim][N] new beliefs (layout per opts)
 *   bw_<type>   [same][dim] their manikde! bandwidths (what setValKDE! stores)
 * Philox streams: convolution row r of a family in iteration i draws stream_offset + (i << 32) + family offset + r (the streams of
 * the device graph's sweep i); the product of the k-th updated variable of a type draws + (3, 4, 6 << 28 for Pose2, Point2, Pose3) + k.
 * N <= 256 (the multiscale Gibbs product). */
enum { ROME_UPSOLVE_SEQUENTIAL = 0, ROME_UPSOLVE_JACOBI = 1 };
typedef struct rome_clique_upsolve_host {
  rome_clique_host clique;
  int32_t gibbs_iters;    /* IIF gibbsIters, default 3 */
  int32_t product_iters;  /* AMP manifoldProduct Niter, default 1 */
  int32_t schedule;       /* ROME_UPSOLVE_* */
  int32_t n_up;
  const int32_t* up_type; const int32_t* up_var;
  int32_t n_msg_pose2, n_msg_point2, n_msg_pose3, reserved0;
  const double* msg_pose2; const int32_t* msg_pose2_up;
  const double* msg_point2; const int32_t* msg_point2_up;
  const double* msg_pose3; const int32_t* msg_pose3_up;
  double* new_pose2; double* bw_pose2;
  double* new_point2; double* bw_point2;
  double* new_pose3; double* bw_pose3;
  /* optional: update GROUPS [n_up], nondecreasing.  Variables of one group are updated together (from the beliefs left by the previous
   * groups), groups one after the other -- replaces `schedule` when given.  A FRONTIER of independent cliques (no frontal of one is a
   * variable of another's factors as anything but a read-only separator) goes through ONE call this way: group g = the g-th frontal of
   * every clique, so that every launch covers all cliques of the frontier ("cliques on the current Bayes-tree frontier
   * are independent").  NULL: groups follow `schedule` (one variable per group / one group). */
  const int32_t* up_group;
  /* optional [n_up]: Philox stream id of the product of updated variable k (NULL: its position among the updated variables of its
   * type): the product draws stream_offset + (it << 32) + (3, 4, 6 << 28) + up_stream[k] -- e.g. the variable's global id, so that a
   * frontier dealt to several ranks draws what the single call draws. */
  const int32_t* up_stream;
  /* optional [n_up], rome_upsolve_plan only: block of the plan's mirror buffer (rome_upsolve_plan_run `mirror_out`) that the new belief
   * of updated variable k is ALSO written to by the product kernel itself, -1 = none: new frontal beliefs land straight in an RCCL send
   * buffer (no gather kernel, no host copy). */
  const int32_t* up_mirror;
  /* optional, rome_upsolve_plan only: messages that LIVE IN THE STORE (the device-resident form of msg_<type>): message m of a type is
   * the belief block smsg_<type>_src[m] of the plan's store AS IT IS WHEN A RUN STARTS -- e.g. the separator belief a child clique's
   * up-solve wrote one tree level earlier (IIF: the TreeBelief a child put!s on its up-message channel) -- and enters
   * every product of the updated variable at position smsg_<type>_up[m] behind the factor proposals and the host messages; its
   * manikde! bandwidths are computed at the start of every run.  The source block must not be a variable the plan updates. */
  int32_t n_smsg_pose2, n_smsg_point2, n_smsg_pose3, reserved1;
  const int32_t* smsg_pose2_src; const int32_t* smsg_pose2_up;
  const int32_t* smsg_point2_src; const int32_t* smsg_point2_up;
  const int32_t* smsg_pose3_src; const int32_t* smsg_pose3_up;
} rome_clique_upsolve_host;
int rome_clique_upsolve(rome_ctx*, const rome_opts*, const rome_clique_upsolve_host*);

/* ---------------------------------------------------------------------------------------------
 * Device-resident clique up-solves: a belief STORE that lives in HBM across calls, and up-solve PLANS over it.
 * rome_clique_upsolve moves every belief of the clique over PCIe on every call; a tree solve visits thousands of cliques and
 * a frontier of independent cliques per tree level, and the separator beliefs one frontier writes are what the
 * next one reads.  With a store the beliefs of the whole graph cross PCIe ONCE; a plan holds the validated row tables of one
 * clique / frontier on the device, and running it issues only kernel launches on the context's stream (no copy, no
 * synchronisation): gibbs_iters x {one convolution launch per factor family -> manikde! bandwidths -> multiscale Gibbs product
 * writing the new beliefs IN PLACE into the store (and, optionally, into an RCCL send buffer)}.
 *   rome_store_create      device blocks [n][dim][N] per variable type (SoA), zero-initialised
 *   rome_store_wrap        the same over caller-owned device memory (e.g. a torch tensor): NULL for an empty type
 *   rome_store_upload / _download   `count` beliefs of one type from `first`, host layout per `layout` (ROME_LAYOUT_*)
 *   rome_store_ptr         device pointer of a type's blocks (for collectives that land in the store)
 * Replaces: the setValKDE! / getBelief traffic around IIF upGibbsCliqueDensity inside solveTree!
 * (src/services/AdditionalUtils.jl:18-19, examples/ManhattanDatasetBatch.jl:43).                                               */
typedef struct rome_store rome_store;
int  rome_store_create(rome_ctx*, int32_t n_particles, int32_t n_pose2, int32_t n_point2, int32_t n_pose3, rome_store** out);
int  rome_store_wrap(rome_ctx*, int32_t n_particles, int32_t n_pose2, double* dev_pose2, int32_t n_point2, double* dev_point2,
                     int32_t n_pose3, double* dev_pose3, rome_store** out);
void rome_store_destroy(rome_store*);
int  rome_store_upload(rome_store*, int32_t layout, int32_t type, int32_t first, int32_t count, const double* host);
int  rome_store_download(rome_store*, int32_t layout, int32_t type, int32_t first, int32_t count, double* host);
int  rome_store_ptr(rome_store*, int32_t type, void** dev, int32_t* n_blocks);
/* A plan = one rome_clique_upsolve_host description bound to a store: `clique.bel_*` and `clique.out_*` are ignored (the variable
 * indices of the row tables, of up_var and of the *_alt columns address the STORE; clique.n_* must not exceed the store's counts),
 * msg_* densities are copied at creation.  new_* / bw_* may be NULL: nothing is downloaded and rome_upsolve_plan_run returns without
 * synchronising (the device-resident mode); when given, the new beliefs / manikde! bandwidths are downloaded at the end of every
 * run.  opts at creation fix n_particles and the host layout of new_* / msg_*; opts at run time give solver, seed, stream_offset,
 * inflation etc. (n_particles must match).
 * rome_upsolve_plan_run(plan, opts, mirror_out, mirror_stride): mirror_out = DEVICE buffer the up_mirror blocks are written to, block
 * m at mirror_out + m * mirror_stride doubles (stride >= N; 0 = 6 N); NULL when the plan has no mirrors.  A block occupies dim * N doubles from
 * its slot: with a stride below 6 N the CALLER lays the slots out so that blocks do not overlap -- e.g. stride N with a Pose2 block taking 3
 * slots, a Point2 block 2 and a Pose3 block 6: an exchange buffer without padding (rome_scatter_plan reads the same layout).
 * Stream semantics: everything a run issues is ordered after the work already queued on the context's stream, and work queued there
 * afterwards is ordered after the run -- inside a run, independent launch chains of a step (per row family: convolutions -> bandwidths;
 * per variable type: ball trees -> product) go to context-owned side streams that fork from and re-join the context's stream. */
typedef struct rome_upsolve_plan rome_upsolve_plan;
int  rome_upsolve_plan_create(rome_ctx*, rome_store*, const rome_opts*, const rome_clique_upsolve_host*, rome_upsolve_plan** out);
int  rome_upsolve_plan_run(rome_upsolve_plan*, const rome_opts*, double* mirror_out, int64_t mirror_stride);
void rome_upsolve_plan_destroy(rome_upsolve_plan*);
/* Block operations inside a store, as plans (index lists uploaded once, a run is ONE launch on the context's stream):
 *   ROME_BLOCKOP_COPY      block dst[k] <- block a[k]                                    (same type)
 *   ROME_BLOCKOP_ANCHOR    block dst[k] <- N copies of ONE point of belief a[k]: the mean (Pose2: circular mean heading; Pose3: mean
 *                          translation, rotation of particle 0) -- the anchor of a relative message: a clique conditions on its anchor
 *                          separator being exactly there
 *   ROME_BLOCKOP_RELATIVE  ref = particle 0 of POSE2 block a[k] (an ANCHOR block).  type[k] = 0: Pose2 block dst[k] <- the tangent
 *                          coordinates of ref^-1 * s_i for the N particles s_i of Pose2 block b[k] (what a p2p2_meas row consumes);
 *                          type[k] = 1: Point2 block dst[k] <- (bearing, range) of the N landmarks of Point2 block b[k] seen from ref
 *                          (what a br1_meas / br0_meas row consumes)
 *   ROME_BLOCKOP_COMPOSE   Pose2 blocks of relative-pose samples, particle by particle: dst[k]_i <- A'_i (+) B'_i, A' = a[k] or its
 *                          inverse (type[k] |= ROME_BLOCKOP_INVERT_A), B' likewise.  Eliminating a variable v of a pose graph with
 *                          sampled edges z_c = v^-1 c, z_k = v^-1 k leaves c^-1 k = z_c^-1 (+) z_k: the pair marginal of the
 *                          neighbours, exactly (elimination.py: variable elimination in relative-factor algebra)
 *   ROME_BLOCKOP_MIX       pooling of independent passes: particle i of dst[k] <- particle i of a[k] unless i % p == p - 1, with
 *                          p = type[k] >> 8 >= 1: after pass p wrote dst, a = the pool of the p - 1 passes before it -- dst becomes a
 *                          mixture in which every pass holds ~N / p particles (its mean: the running average of the passes)
 * type[k] = variable type of entry k (0 Pose2 / 1 Point2 / 2 Pose3; RELATIVE: of b and dst); b may be NULL except for RELATIVE / COMPOSE.
 * Replaces (with smsg_* and p2p2_meas): the up-message channels between cliques of IIF's solveTree! (put!/take! of
 * TreeBelief messages), kept device-resident. */
enum { ROME_BLOCKOP_COPY = 0, ROME_BLOCKOP_ANCHOR = 1, ROME_BLOCKOP_RELATIVE = 2, ROME_BLOCKOP_COMPOSE = 3, ROME_BLOCKOP_MIX = 4 };
#define ROME_BLOCKOP_INVERT_A 0x100   /* COMPOSE: OR into type[k] -- take the inverse of every particle of block a[k] ... */
#define ROME_BLOCKOP_INVERT_B 0x200   /* ... of block b[k] */
typedef struct rome_blockop_plan rome_blockop_plan;
int  rome_blockop_plan_create(rome_ctx*, rome_store*, int32_t op, int32_t n, const int32_t* type, const int32_t* a, const int32_t* b,
                              const int32_t* dst, rome_blockop_plan** out);
/* COMPOSE with per-entry inflation: params[2k], params[2k + 1] > 0 scale the deviations of entry k's composed samples about their mean
 * (translation, heading) -- the star-mesh transform of an eliminated star: the edge between two of its legs keeps the composed mean and takes
 * the variance v_j + v_k + v_j v_k sum_{i != j,k} 1 / v_i.  params = NULL: plain compositions (rome_blockop_plan_create). */
int  rome_blockop_plan_create_ex(rome_ctx*, rome_store*, int32_t op, int32_t n, const int32_t* type, const int32_t* a, const int32_t* b,
                                 const int32_t* dst, const double* params, rome_blockop_plan** out);
int  rome_blockop_plan_run(rome_blockop_plan*);
void rome_blockop_plan_destroy(rome_blockop_plan*);
/* A scatter plan: the receive side of a frontier exchange.  After an all-gather of the ranks' send buffers, block src_block[k] of the
 * receive buffer (units of `stride` doubles, stride >= N, 0 = 6 N) is the new belief of variable (type[k], var[k]) of the store; the lists are
 * uploaded once, a run is ONE launch on the context's stream. */
typedef struct rome_scatter_plan rome_scatter_plan;
int  rome_scatter_plan_create(rome_ctx*, rome_store*, int32_t n, const int32_t* type, const int32_t* var, const int32_t* src_block,
                              int64_t stride, rome_scatter_plan** out);
int  rome_scatter_plan_run(rome_scatter_plan*, const double* src_dev);
void rome_scatter_plan_destroy(rome_scatter_plan*);

/* ---------------------------------------------------------------------------------------------
 * Graph-indexed DEVICE-pointer variant: beliefs stay resident in HBM (SoA blocks [var][dim][N]),
 * one launch sweeps a whole table of (factor, direction) convolutions.  This is what a clique /
 * whole-graph sweep of `solveTree!` (examples/ManhattanDatasetBatch.jl:43) issues.
 * All pointers below are device pointers valid on the context's device.                          */
typedef struct rome_conv_dev {
  int32_t n_conv;            /* C */
  int32_t dir_all;           /* used when dir == NULL (bearing-range: always)                     */
  const int32_t* factor;     /* [C] row of mu/L, NULL -> c                                        */
  const int32_t* dir;        /* [C], NULL -> dir_all                                              */
  const int32_t* fixed_var;  /* [C] block index into bel_fixed, NULL -> c                         */
  const int32_t* target_var; /* [C] block index into bel_target, NULL -> c                        */
  const double* mu;          /* [F][dz]                                                           */
  const double* L;           /* [F][dz(dz+1)/2] packed lower Cholesky (bearing-range: [F][2] sigmas) */
  const double* bel_fixed;   /* blocks of the fixed variable type                                 */
  const double* bel_target;  /* blocks of the target variable type (start points)                 */
  const double* noise;       /* [C][dz][N] or NULL                                                */
  double* out;               /* [C][dt][N] proposals                                              */
  int32_t* status;           /* [C][N] or NULL                                                    */
  /* optional: up to 4 convolution rows (more: mirror_map below) whose proposal block is ALSO written to mirror_out[m] ([dt][N] each),
   * e.g. separator beliefs straight into an RCCL send buffer (no gather kernel between sweep and collective) */
  int32_t n_mirror;
  int32_t mirror_row[4];
  int32_t reserved;
  double* mirror_out;
  /* optional, bearing-range only: `multihypo=[1, w, 1-w]` over two landmark candidates (IIF addFactor! kwarg,
   * test/testMultimodalRangeBearing.jl:53).  alt_var[c] = block of the OTHER landmark (-1: ordinary row),
   * hypo_w[c] = probability that the row's own landmark (fixed_var for dir 1, target_var for dir 0) is the sighted one. */
  const int32_t* alt_var;
  const double* hypo_w;
  /* optional: IIF `nullhypo=p` per row (addFactor!(fg, [:x3;:x1], odoc3, nullhypo=0.5), test/testPose3Pose3NH.jl:118):
   * with probability nullhypo[c] a particle is not constrained by the factor: it keeps its start value and gets
   * spread_nh · std entropy (std = root of the Fréchet variance of the target's start belief).  NULL -> 0 everywhere. */
  const double* nullhypo;
  /* optional: the four table columns interleaved, [C][4] int32 = (factor, dir, fixed_var, target_var) per row.  When given it
   * REPLACES the column pointers above (which may then be NULL) and, for rows without pre-sampled noise / multihypo /
   * nullhypo, selects the plain sweep kernels: the packed sweep (k_conv_flat: thread = two neighbouring particles, 5 rows per
   * 256-thread block at N = 100, per-factor constants staged through LDS) for the unique-root factors under CLOSED_FORM / NEWTON, the
   * lean wave-per-row kernel otherwise.  Bearing-range rows ignore the dir entry. */
  const int32_t* rows4;
  /* optional: ANY number of mirrored rows (replaces n_mirror / mirror_row when given): mirror_map[c] = block of mirror_out that
   * row c's proposal is also written to, -1 = none.  A Bayes-tree cut / the beehive lattice publishes more than four separators. */
  const int32_t* mirror_map;
} rome_conv_dev;

int rome_conv_pose2pose2_dev(rome_ctx*, const rome_opts*, const rome_conv_dev*);
int rome_conv_pose2point2br_dev(rome_ctx*, const rome_opts*, const rome_conv_dev*);
int rome_conv_pose3pose3_dev(rome_ctx*, const rome_opts*, const rome_conv_dev*);
/* The whole convolution sweep of a Pose2 / Point2 graph (odometry + bearing-range sightings: MIT.g2o with landmarks, the beehive) in ONE
 * call: the Pose2Pose2 (+ PriorPose2 rows) table, the bearing-range -> pose table (dir_all = 1) and the bearing-range -> landmark table
 * (dir_all = 0); any of them may be NULL.  Philox stream of row r of family k = opts->stream_offset + family_stream_offset[k] + r
 * (k = 0 p2p2, 1 br1, 2 br0; NULL = no family offsets).  When every table takes its plain kernel (rows4 given, in-kernel noise, no
 * multihypo / nullhypo / status, CLOSED_FORM, NEWTON or GAUSS_NEWTON, 64 < N <= 128) the three families run as ONE fused launch -- same proposals
 * bit for bit as three rome_conv_*_dev calls, which is what happens otherwise. */
int rome_sweep_pose2_dev(rome_ctx*, const rome_opts*, const rome_conv_dev* p2p2, const rome_conv_dev* br1, const rome_conv_dev* br0,
                         const uint64_t* family_stream_offset /*[3] or NULL*/);
int rome_sample_priorpose2_dev(rome_ctx*, const rome_opts*, const rome_conv_dev*); /* uses factor, mu, L, noise, out */
int rome_sample_priorpose3_dev(rome_ctx*, const rome_opts*, const rome_conv_dev*);
int rome_sample_priorpoint2_dev(rome_ctx*, const rome_opts*, const rome_conv_dev*);   /* L = [F][3] packed Cholesky of the 2x2 covariances */

/* ---------------------------------------------------------------------------------------------
 * Parametric path: batched whitened residuals + analytic Jacobians at the
 * measurement mean.  Replaces the per-factor residual/Jacobian evaluation IIF.solveGraphParametric!
 * does through the same RoME functors (call sites src/services/AdditionalUtils.jl:22,37;
 * getMeasurementParametric src/factors/BearingRange2D.jl:30-37):  cost = Σ_f ‖W_f r_f(μ_f; x)‖², WᵀW = Σ⁻¹.
 * Rows are per factor: mu [F][dz], W [F][dr*dr] row-major whitening matrix, xa/xb [F][da|db] the
 * coordinates of the factor's 1st/2nd variable (gathered by the caller), outputs r [F][dr],
 * Ja [F][dr*da], Jb [F][dr*db] (row-major; Jb/xb NULL for priors).  Perturbations: Pose2 (δx,δy,δθ)
 * with x⊕δ = ((t+δt), R(θ+δθ)); Point2 additive; Pose3 (δt, δω) with x⊕δ = ((t+δt), R·Exp(δω)).
 *   kind            dz dr da db   residual (reference file:line)
 *   PRIORPOSE2       3  3  3  -   src/factors/PriorPose2.jl:37-47
 *   POSE2POSE2       3  3  3  3   src/factors/Pose2D.jl:51-67
 *   POSE2POINT2BR    2  2  3  2   src/factors/BearingRange2D.jl:48-64
 *   PRIORPOINT2      2  2  2  -   src/factors/Point2D.jl:14-18
 *   POSE3POSE3       6  6  6  6   src/factors/Pose3Pose3.jl:17-29
 *   PRIORPOSE3       6  6  6  -   src/factors/Pose3D.jl:15-19                                     */
enum { ROME_FACTOR_PRIORPOSE2 = 0, ROME_FACTOR_POSE2POSE2 = 1, ROME_FACTOR_POSE2POINT2BR = 2,
       ROME_FACTOR_PRIORPOINT2 = 3, ROME_FACTOR_POSE3POSE3 = 4, ROME_FACTOR_PRIORPOSE3 = 5 };
int rome_linearize(rome_ctx*, int32_t kind, int32_t F, const double* mu, const double* W,
                   const double* xa, const double* xb, double* r, double* Ja, double* Jb);      /* host pointers   */
int rome_linearize_dev(rome_ctx*, int32_t kind, int32_t F, const double* mu, const double* W,
                       const double* xa, const double* xb, double* r, double* Ja, double* Jb);  /* device pointers */

/* ---------------------------------------------------------------------------------------------
 * Belief summaries and proposal product, DEVICE pointers, SoA blocks.
 * rome_belief_stats*: manifold mean and per-coordinate std of V beliefs ([V][dim][N]; dim 2 Point2, 3 Pose2,
 *   6 Pose3) -> mean [V][dim], std [V][dim].  Replaces `mean(M, pts)` / `std(vartype, pts)` as used for PPEs
 *   (examples/ManhattanBatchAnalysis.jl:59-62) and for IIF's inflation spread (calcStdBasicSpread).
 * rome_product_dev: new belief of every variable from the proposals that target it (CSR prop_ptr[V+1] /
 *   prop_rows into prop [rows][dim][N]); variables without proposals keep bel_in.  STAND-IN for
 *   AMP.manifoldProduct (unvendored): importance-sampling product of the proposal KDEs, see DESIGN.md §11.
 *   dim 2 (Point2), 3 (Pose2) or 6 (Pose3, coordinates [t; rotation vector], N <= 256). */
int rome_belief_stats_dev(rome_ctx*, int32_t dim, int32_t V, int32_t N, const double* bel, double* mean, double* std);
int rome_belief_stats(rome_ctx*, int32_t dim, int32_t V, int32_t N, const double* bel, double* mean, double* std); /* host pointers */
/* rome_kde_bandwidth*: the bandwidth `manikde!` selects for a belief, one per coordinate, by leave-one-out likelihood
 *   cross-validation of a 1-D Gaussian KDE (golden section on the bracket of KDE.jl's ksize "lcv"; coordinate k is circular
 *   -- differences wrapped -- when bit k of circular_mask is set: Pose2 = 0b100).  bel [V][dim][N] -> bw [V][dim], dim 1..6,
 *   2 <= N <= ROME_MAX_PARTICLES.  tol_* <= 0 selects the reference's stopping rules (1e-2 relative for Euclidean
 *   coordinates, 1e-6 for circular ones).  Reproduces the bandwidths stored in the reference's solved graph
 *   (examples/fg-after-solve.tar.gz `vecbw`; tests/test_gpu_kde.py). */
int rome_kde_bandwidth_dev(rome_ctx*, int32_t dim, int32_t V, int32_t N, const double* bel, uint32_t circular_mask,
                           double tol_euclid, double tol_circular, double* bw);
int rome_kde_bandwidth(rome_ctx*, int32_t dim, int32_t V, int32_t N, const double* bel, uint32_t circular_mask,
                       double tol_euclid, double tol_circular, double* bw);                                        /* host pointers */
/* rome_kde_max*: max-density point estimate of a belief, coordinate by coordinate -- IIF's getKDEMax, the `max` entry of a variable's
 *   PPE (and the heading of `suggested` in the stored graph): the Euclidean marginal KDE with bandwidth bw [V][dim] is evaluated on
 *   grid_points (<= 0: the reference's 200; max 256) equispaced points over the particle range extended by 10 % on both sides; the first
 *   maximiser is returned.  out [V][dim].  With the stored bandwidths this reproduces every `ppe.max` of examples/fg-after-solve.tar.gz. */
int rome_kde_max_dev(rome_ctx*, int32_t dim, int32_t V, int32_t N, const double* bel, const double* bw, int32_t grid_points, double* out);
int rome_kde_max(rome_ctx*, int32_t dim, int32_t V, int32_t N, const double* bel, const double* bw, int32_t grid_points, double* out); /* host */
int rome_product_dev(rome_ctx*, const rome_opts*, int32_t dim, int32_t V, const int32_t* prop_ptr, const int32_t* prop_rows,
                     const double* prop, const double* bel_in, double* bel_out);
/* same with caller-supplied kernel bandwidths of the proposals, prop_bw [rows][dim] (e.g. rome_kde_bandwidth_dev run on `prop`,
 * which is what the reference's manikde! attaches to every convolution result); NULL = Silverman's rule in-kernel. */
int rome_product_bw_dev(rome_ctx*, const rome_opts*, int32_t dim, int32_t V, const int32_t* prop_ptr, const int32_t* prop_rows,
                        const double* prop, const double* prop_bw, const double* bel_in, double* bel_out);
/* rome_product_gibbs_dev: the reference's own product -- ⚠AMP `manifoldProduct(ff, manifold; Niter)` -> ⚠KDE.jl `prodAppxMSGibbsS`:
 *   multiscale Gibbs sampling from the product of the proposal KDEs (Ihler, Sudderth, Freeman, Willsky, NIPS 2003), restated from
 *   the paper (neither package is vendored: statistical pins only; oracle/rome_oracle.c ro_product_msgibbs is the step-by-step
 *   definition the kernel is compared with).  Same CSR layout as rome_product_dev; prop_bw [rows][dim] = the bandwidth manikde!
 *   attached to every proposal (rome_kde_bandwidth_dev) is REQUIRED; coordinate k is circular when bit k of circular_mask is set
 *   (Pose2: 0b100); gibbs_iters = AMP's Niter (1); max_proposals >= max_v (prop_ptr[v+1] - prop_ptr[v]) sizes the LDS of a block
 *   (the CSR is built on the host, so the caller knows it); n_prop_rows = rows of prop / prop_bw (a ball tree is built for each, in a
 *   context-owned workspace of 6.8 kB per row for Pose2).  dim 2 (Point2), 3 (Pose2) or 6 (Pose3: coordinates [t; rotation vector];
 *   the rotation coordinates of a proposal live in the chart at the rotation of its point 0 -- Log(R_0ᵀ R_i) -- where the tree and
 *   every candidate evaluation are Euclidean; only the product Gaussians of selected nodes change charts, by Exp / Log; pass
 *   circular_mask = 0), N <= 256.  Variables without proposals keep bel_in, with one proposal take it unchanged (as AMP does). */
int rome_product_gibbs_dev(rome_ctx*, const rome_opts*, int32_t dim, int32_t V, const int32_t* prop_ptr, const int32_t* prop_rows,
                           const double* prop, const double* prop_bw, int32_t n_prop_rows, const double* bel_in, double* bel_out,
                           uint32_t circular_mask, int32_t gibbs_iters, int32_t max_proposals);

/* thin device-memory helpers for callers without their own HIP runtime binding (e.g. the Julia shim) */
int rome_dev_alloc(rome_ctx*, uint64_t bytes, void** out);
int rome_dev_free(rome_ctx*, void* p);
int rome_dev_upload(rome_ctx*, void* dst_dev, const void* src_host, uint64_t bytes);
int rome_dev_download(rome_ctx*, void* dst_host, const void* src_dev, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* ROME_MI355_H */
