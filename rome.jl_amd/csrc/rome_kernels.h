// rome_kernels.h -- internal interface between the C-ABI host layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rome {

enum { kSolverClosedForm = 0, kSolverNewton = 1, kSolverNelderMead = 2, kSolverGaussNewton = 3 };
enum { kDirTo = 0, kDirFrom = 1, kDirPrior = 2 };  // kDirPrior: row is a prior (no fixed variable): proposal = sample

// All pointers are DEVICE pointers.  Belief / proposal blocks are SoA: [block][dim][N].
struct ConvArgs {
  int n_conv;                 // C convolutions (= waves)
  int N;                      // particles per belief
  const int32_t* factor;      // [C] row of mu/L, or nullptr (identity)
  const int32_t* dir;         // [C] 0: solve 2nd ("to") variable, 1: solve 1st ("from"); nullptr -> dir_all
  const int32_t* fixed_var;   // [C] block of bel_fixed, or nullptr (identity)
  const int32_t* target_var;  // [C] block of bel_target (start points u0), or nullptr (identity)
  const int32_t* rows4;       // [C][4] (factor, dir, fixed_var, target_var) interleaved, or nullptr; replaces the four columns
  const double* mu;           // [F][dz]
  const double* L;            // [F][dz(dz+1)/2] row-packed lower Cholesky of Σ  (bearing-range: [F][2] sigmas)
  const double* bel_fixed;
  const double* bel_target;
  const double* noise;        // [C][dz][N] standard normals (or, noise_is_meas: the measurement samples themselves), or nullptr -> in-kernel Philox
  int noise_is_meas;
  double* out;                // [C][dt][N]
  int32_t* status;            // [C][N] or nullptr
  const int32_t* alt_var;     // [C] bearing-range multihypo: the other landmark candidate (-1: single hypothesis), or nullptr
  const double* hypo_w;       // [C] probability that the table's own landmark is the sighted one
  double spread_nh;           // IIF SolverParams.spreadNH
  const double* nullhypo;     // [C] probability that the factor does not apply to a particle (IIF nullhypo=), or nullptr
  int n_mirror;               // rows additionally written to mirror_out[m] (separator beliefs -> send buffer)
  int mirror_row[4];
  double* mirror_out;
  const int32_t* mirror_map;  // [C] block of mirror_out row c is ALSO written to (-1: none), or nullptr -> mirror_row; any number of rows
  const int32_t* meas_block;  // [C] block of meas_base ([.][dz][N]) holding row c's MEASUREMENT samples (-1: sample the factor in-kernel), or
                              // nullptr: a factor whose measurement distribution is a set of samples -- the relative message of a child clique
  const double* meas_base;
  const int32_t* row_stream;  // [C] Philox stream id of row c (stream = stream_offset + id), or nullptr -> c.  Served by the wave-per-row
                              // kernels and the prior samplers (a table with this column does not take the packed sweep)
  int dir_all;
  int max_iters;
  int cycles;
  double tol;
  double inflation;
  double inv_n;               // 1/N and 1/(N-1) (host-computed: no FP64 divide in the kernel)
  double inv_nm1;
  uint64_t seed;
  uint64_t stream_offset;
};

hipError_t launch_conv_pose2pose2(const ConvArgs& a, int solver, hipStream_t s);
hipError_t launch_conv_bearingrange(const ConvArgs& a, int solver, hipStream_t s);
hipError_t launch_conv_pose3pose3(const ConvArgs& a, int solver, hipStream_t s);
hipError_t launch_sweep_pose2(const ConvArgs* p2p2, const ConvArgs* br1, const ConvArgs* br0, int solver, hipStream_t s);
hipError_t launch_sample_priorpose2(const ConvArgs& a, hipStream_t s);
hipError_t launch_sample_priorpose3(const ConvArgs& a, hipStream_t s);
hipError_t launch_sample_priorpoint2(const ConvArgs& a, hipStream_t s);

hipError_t launch_residual_pose2pose2(int n, const double* z, const double* p, const double* q, double* r, hipStream_t s);
hipError_t launch_residual_priorpose2(int n, const double* m, const double* p, double* r, hipStream_t s);
hipError_t launch_residual_bearingrange(int n, const double* z, const double* p, int p_is_point, const double* l, double* r, hipStream_t s);
hipError_t launch_residual_pose3pose3(int n, const double* z, const double* p, const double* q, int pts, double* r, hipStream_t s);
hipError_t launch_residual_priorpose3(int n, const double* m, const double* p, double* r, hipStream_t s);

hipError_t launch_points_to_coords(int n, int dim, const double* pts, double* c, hipStream_t s);
hipError_t launch_coords_to_points(int n, int dim, const double* c, double* pts, hipStream_t s);
hipError_t launch_linearize(int kind, int F, const double* mu, const double* W, const double* xa, const double* xb,
                            double* r, double* Ja, double* Jb, hipStream_t s);

hipError_t launch_belief_stats(int dim, int V, int N, const double* bel, double* mean, double* sdev, hipStream_t s);
// block_idx: optional [V] block of `bel` that belief v lives in (nullptr: v) -- the bandwidths of a scattered subset of a store
hipError_t launch_kde_bandwidth(int dim, int V, int N, const double* bel, uint32_t circ_mask, double tol_e, double tol_c,
                                double* bw, int32_t* evals, hipStream_t s, const int32_t* block_idx = nullptr);
hipError_t launch_kde_max(int dim, int V, int N, int G, double extend, const double* bel, const double* bw, double* out, hipStream_t s);
hipError_t launch_product(int dim, int V, int N, const int32_t* prop_ptr, const int32_t* prop_rows, const double* prop,
                          const double* prop_bw, const double* bel_in, double* bel_out, double c_n, uint64_t seed,
                          uint64_t stream_offset, hipStream_t s);

size_t gibbs_workspace_bytes(int dim, int n_rows, int V, int N);   // (N: 128-slot trees up to N = 128, 256-slot trees above)
// GibbsPlace (optional): where the V variables of a launch live and what they draw -- var_block[v] = block of bel_in / bel_out
// (nullptr: v; bel_in == bel_out is allowed then: a product only reads its own variable's block), var_stream[v] = Philox stream id
// (nullptr: v), mirror_slot[v] = block of mirror_out (stride doubles apart) the new belief is ALSO written to (-1 / nullptr: none)
struct GibbsPlace { const int32_t* var_block; const int32_t* var_stream; const int32_t* mirror_slot; double* mirror_out; int64_t mirror_stride; };
hipError_t launch_product_gibbs(int dim, int V, int N, int n_rows, const int32_t* prop_ptr, const int32_t* prop_rows, const double* prop,
                                const double* prop_bw, const double* bel_in, double* bel_out, void* trees, uint32_t circ, int iters, int max_k,
                                uint64_t seed, uint64_t stream_offset, hipStream_t s, const GibbsPlace* place = nullptr);
// store <-> blocks of a device buffer, one launch: entry k = (dim, var, block, type); to_store: belief `var` of the type's store array
// <- block `block` of buf (stride doubles apart); else the reverse (a contiguous download buffer <- scattered beliefs)
hipError_t launch_scatter_blocks(int n, int N, const int32_t* ent /*[n][4]*/, const double* buf, int64_t stride,
                                 double* st2, double* st_pt, double* st3, hipStream_t s, int to_store);

// block operations inside a store (include/rome_mi355.h ROME_BLOCKOP_*): entry k = (type, a, b, dst); one 256-thread block per entry
hipError_t launch_block_ops(int op, int n, int N, const int32_t* ent /*[n][4]*/, double* st2, double* st_pt, double* st3, hipStream_t s, const double* prm /*[n][2] or NULL*/ = nullptr);

}  // namespace rome
