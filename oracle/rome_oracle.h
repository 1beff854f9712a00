/* rome_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, FP64) of the non-parametric factor-convolution hot
 * path of RoME.jl + IncrementalInference.jl, used as the parity checker for
 * the HIP kernels and as the `cpu_baseline` ("port") leg of bench.py.
 * Nothing under rome.jl_amd/ (the product) may include, link or call this.
 *
 * Citation convention: `path:line` = /root/reference/path:line (RoME.jl v0.24.6).
 * "⚠IIF/⚠Manifolds/⚠Optim" = behaviour of an UNVENDORED Julia dependency
 * (IncrementalInference 0.35, Manifolds 0.10.1, Optim 1.x; Project.toml:50-84),
 * restated from the published algorithm and anchored on RoME's call sites/tests.
 *
 * Parity pin status:
 *   - residual functors: PINNED by the reference's known-answer tests
 *     (tests/golden/residual_kats.json, transcribed from the test .jl files).
 *   - per-particle optimiser trajectory, RNG stream, entropy inflation:
 *     PARITY UNPINNED by the reference (SURVEY.md 8(c) "Unpinned").
 */
#ifndef ROME_ORACLE_H
#define ROME_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { RO_SOLVER_CLOSED_FORM = 0, RO_SOLVER_NEWTON = 1, RO_SOLVER_NELDER_MEAD = 2 };

typedef struct {
  int32_t n_particles;    /* N (IIF default 100, src/canonical/GenerateHexagonal.jl:30)   */
  int32_t solver;         /* RO_SOLVER_*                                                   */
  int32_t max_iters;      /* Newton: 20 ; Nelder-Mead: 1000 (Optim default)                */
  int32_t inflate_cycles; /* IIF SolverParams.inflateCycles = 3                            */
  double  tol;            /* Newton: stop when max|r| <= tol ; NM: g_tol (Optim 1e-8)      */
  double  inflation;      /* IIF SolverParams.inflation (kappa), default 5.0 ; 0 = off     */
  uint64_t seed;          /* Philox key                                                    */
  uint64_t stream_offset; /* Philox stream id = stream_offset + global convolution index   */
  double  nullhypo;       /* IIF nullhypo= probability applied to every row of the call (0 = off) */
  double  spread_nh;      /* IIF spreadNH (3.0)                                              */
} ro_opts;

/* ---- counter-based RNG (shared definition with the HIP path) ---- */
void ro_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void ro_box_muller(uint32_t wa, uint32_t wb, double* n0, double* n1);
void ro_rng_normals(uint64_t seed, uint64_t stream, uint32_t particle, int d, double* out);
void ro_rng_entropy(uint64_t seed, uint64_t stream, uint32_t particle, int cycle, int d, double* out);

/* ---- manifold primitives ---- */
void ro_pose2_point_from_coords(const double c[3], double pt[6]);   /* (x,y,th) -> [tx,ty,R11,R21,R12,R22] */
void ro_pose2_coords_from_point(const double pt[6], double c[3]);
void ro_pose3_point_from_coords(const double c[6], double pt[12]);  /* (t,omega) -> [t(3), R col-major(9)] */
void ro_pose3_coords_from_point(const double pt[12], double c[6]);
void ro_so3_exp(const double w[3], double R[9]);                     /* col-major */
void ro_so3_log(const double R[9], double w[3]);
double ro_sym_rem(double x);
void ro_rotxyz(double roll, double pitch, double yaw, double R[9]);  /* Rotations.jl RotXYZ = Rx*Ry*Rz */

/* ---- residual functors on native points (what the reference's CalcFactor sees) ---- */
void ro_residual_pose2pose2_pt(const double X[6], const double p[6], const double q[6], double r[3]);
void ro_residual_priorpose2_pt(const double m[6], const double p[6], double r[3]);
void ro_residual_pose2point2br_pt(const double meas[5], const double p[6], const double l[2], double r[2]);
void ro_residual_pose3pose3_pt(const double X[12], const double p[12], const double q[12], double r[6]);
void ro_residual_priorpose3_pt(const double m[12], const double p[12], double r[6]);
/* ---- same, batched on coordinates (n rows) ---- */
void ro_residual_pose2pose2(int n, const double* z, const double* p, const double* q, double* r);        /* n x3 each   */
void ro_residual_priorpose2(int n, const double* m, const double* p, double* r);                         /* n x3        */
void ro_residual_pose2point2br(int n, const double* z, const double* p, const double* l, double* r);     /* z nx2 p nx3 l nx2 r nx2 */
void ro_residual_pose3pose3(int n, const double* z, const double* p, const double* q, double* r);        /* n x6        */
void ro_residual_priorpose3(int n, const double* m, const double* p, double* r);

/* ---- small helpers ---- */
int  ro_cholesky_lower(int d, const double* cov /*d x d row-major*/, double* Lpacked /* d(d+1)/2 row-packed lower */);
void ro_belief_spread_se2(int N, const double* x, const double* y, const double* th, double* mean3, double* std3);
void ro_belief_spread_r2(int N, const double* x, const double* y, double* mean2, double* std2);
void ro_belief_spread_se3(int N, const double* blk /*[6][N]*/, double* mean6, double* std6);

/* ---- generic Nelder-Mead exactly as Optim.jl's defaults (⚠Optim) ---- */
typedef double (*ro_cost_fn)(const double* x, void* ctx);
int ro_nelder_mead(int n, ro_cost_fn f, void* ctx, double* x /*in: x0, out: minimiser*/,
                   int max_iters, double g_tol, int* n_evals);

/* ---- factor convolutions: C convolutions x N particles, SoA blocks [blk][d][N] ----
 * factor[c]  : row of mu/L (NULL -> c)           dir[c] : 0 solve the "to"/second variable, 1 solve the "from"/first
 * fixed_var[c], target_var[c]: block index into the belief arrays (NULL -> c)
 * noise: standard-normal xi, [C][dz][N], or NULL -> Philox (ro_rng_normals)
 * out  : [C][dt][N] ; status: [C][N] (0 converged, 1 = iteration cap) or NULL                                    */
int ro_conv_pose2pose2(const ro_opts* o, int C, const int32_t* factor, const int32_t* dir,
                       const int32_t* fixed_var, const int32_t* target_var,
                       const double* mu /*[F][3]*/, const double* L /*[F][6]*/,
                       const double* bel /*[V][3][N]*/, const double* noise, double* out, int32_t* status);
int ro_conv_pose2pose2_mh(const ro_opts* o, int C, const int32_t* factor, const int32_t* dir,
                       const int32_t* fixed_var, const int32_t* target_var,
                       const double* mu /*[F][3]*/, const double* L /*[F][6]*/,
                       const double* bel /*[V][3][N]*/, const double* noise, double* out, int32_t* status,
                          const int32_t* alt_var, const double* hypo_w, double spread_nh);
int ro_conv_pose2point2br(const ro_opts* o, int C, const int32_t* factor, int dir,
                          const int32_t* fixed_var, const int32_t* target_var,
                          const double* mu /*[F][2] (bearing,range)*/, const double* sigma /*[F][2]*/,
                          const double* bel_fixed /*dir0: poses [V][3][N]; dir1: points [V][2][N]*/,
                          const double* bel_target /*dir0: points; dir1: poses*/,
                          const double* noise, double* out, int32_t* status);
int ro_conv_pose2point2br_mh(const ro_opts* o, int C, const int32_t* factor, int dir,
                             const int32_t* fixed_var, const int32_t* target_var,
                             const double* mu, const double* sigma,
                             const double* bel_fixed, const double* bel_target,
                             const double* noise, double* out, int32_t* status,
                             const int32_t* alt_var /*[C] or NULL*/, const double* hypo_w /*[C]*/, double spread_nh);
int ro_sample_priorpose2(const ro_opts* o, int C, const int32_t* factor,
                         const double* mu, const double* L, const double* noise, double* out /*[C][3][N]*/);
int ro_sample_priorpoint2(const ro_opts* o, int C, const int32_t* factor,
                          const double* mu /*[F][2]*/, const double* L /*[F][3]*/, const double* noise, double* out /*[C][2][N]*/);
int ro_conv_pose3pose3(const ro_opts* o, int C, const int32_t* factor, const int32_t* dir,
                       const int32_t* fixed_var, const int32_t* target_var,
                       const double* mu /*[F][6]*/, const double* L /*[F][21]*/,
                       const double* bel /*[V][6][N]*/, const double* noise, double* out, int32_t* status);
int ro_sample_priorpose3(const ro_opts* o, int C, const int32_t* factor,
                         const double* mu, const double* L, const double* noise, double* out /*[C][6][N]*/);

/* product of K proposal beliefs per variable (stand-in for AMP manifoldProduct; see rome_oracle.c) */
/* leave-one-out likelihood bandwidths (manikde! rule; pinned on the reference's stored bandwidths) */
double ro_kde_bandwidth_lcv(int N, const double* x, int circular, double tol, int* n_evals);
int ro_kde_bandwidths(int dim, int V, int N, const double* bel /*[V][dim][N]*/, uint32_t circular_mask, double tol_euclid,
                      double tol_circular, double* bw /*[V][dim]*/);
/* per-coordinate max-density point on the reference's 200-point grid (IIF getKDEMax; pinned on the stored ppe.max) */
int ro_kde_max(int dim, int V, int N, const double* bel, const double* bw, int G, double extend, double* out);
int ro_product(const ro_opts* o, int dim, int V, const int32_t* prop_ptr /*[V+1]*/, const int32_t* prop_rows,
               const double* prop /*[rows][dim][N]*/, const double* bel_in /*[V][dim][N]*/, double* bel_out);
/* ⚠AMP manifoldProduct restated: multiscale Gibbs sampling from the product of the proposal KDEs (see rome_oracle.c);
 * dim 2 (Point2) / 3 (Pose2, circular_mask 0b100); prop_bw [rows][dim] = the bandwidths manikde! attached to every proposal. */
int ro_product_msgibbs(const ro_opts* o, int dim, int V, const int32_t* prop_ptr, const int32_t* prop_rows, const double* prop,
                       const double* prop_bw, const double* bel_in, uint32_t circular_mask, int gibbs_iters, double* bel_out);
int ro_product_bw(const ro_opts* o, int dim, int V, const int32_t* prop_ptr, const int32_t* prop_rows, const double* prop,
                  const double* prop_bw /*[rows][dim] or NULL*/, const double* bel_in, double* bel_out);

int ro_num_threads(void);
void ro_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
