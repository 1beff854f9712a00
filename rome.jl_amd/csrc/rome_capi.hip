// rome_capi.hip -- extern "C" boundary of librome_mi355.so (see include/rome_mi355.h).
// Host-side plumbing only: argument checks, Cholesky of the measurement covariances, layout
// conversion + staging for the host-pointer entry points, kernel launches.  No CPU compute path.
#include "../../include/rome_mi355.h"
#include "rome_kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

struct rome_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipError_t last_hip = hipSuccess;
  static constexpr int kBufs = 13;   // 9 clique arena, 10 Gibbs trees, 11 / 12 temporary store / plan of the one-shot up-solve
  void* dbuf[kBufs] = {nullptr};
  size_t dcap[kBufs] = {0};
  // pinned host staging of the host-pointer entry points (layout conversion writes straight into DMA-able memory)
  static constexpr int kHostBufs = 4;   // 0 fixed, 1 target (+ alternative landmark blocks), 2 noise, 3 out
  void* hbuf[kHostBufs] = {nullptr};
  size_t hcap[kHostBufs] = {0};
  // fork / join inside an up-solve step (plan_run): independent launches of a step -- the row families' convolutions + bandwidths, then
  // the products of the variable types -- go to side streams and re-join `stream`; created on first use
  static constexpr int kSide = 5;
  hipStream_t side[kSide] = {nullptr};
  hipEvent_t ev_fork = nullptr, ev_side[kSide] = {nullptr}, ev_side2[kSide] = {nullptr};   // two event sets: the phases alternate
  hipEvent_t ev_order = nullptr;   // rome_ctx_set_stream: the new stream is ordered after everything queued on the previous one
};

namespace {

inline int hip_fail(rome_ctx* c, hipError_t e) {
  if (c) c->last_hip = e;
  return ROME_ERR_HIP;
}
#define ROME_HIP(ctx, expr)                                   \
  do {                                                        \
    hipError_t _e = (expr);                                   \
    if (_e != hipSuccess) return hip_fail((ctx), _e);         \
  } while (0)

// Every entry point that launches or copies binds the thread to the context's device first (a context created for device k
// must work whatever the caller's current device is); one hipGetDevice when it already is current.
#define ROME_BIND(ctx)                                                   \
  do {                                                                   \
    int _cur = -1;                                                       \
    if (hipGetDevice(&_cur) != hipSuccess || _cur != (ctx)->device)      \
      ROME_HIP((ctx), hipSetDevice((ctx)->device));                      \
  } while (0)

int ensure(rome_ctx* c, int idx, size_t bytes, void** out) {
  if (bytes == 0) bytes = 8;
  if (c->dcap[idx] < bytes) {
    if (c->dbuf[idx]) { hipError_t e = hipFree(c->dbuf[idx]); if (e != hipSuccess) return hip_fail(c, e); c->dbuf[idx] = nullptr; c->dcap[idx] = 0; }
    size_t cap = bytes + bytes / 4;
    hipError_t e = hipMalloc(&c->dbuf[idx], cap);
    if (e != hipSuccess) return hip_fail(c, e);
    c->dcap[idx] = cap;
  }
  *out = c->dbuf[idx];
  return ROME_OK;
}

int ensure_side(rome_ctx* c) {
  if (c->ev_fork) return ROME_OK;
  for (int i = 0; i < rome_ctx::kSide; ++i) {   // (a failure half way leaves what exists for the next attempt and for rome_ctx_destroy)
    if (!c->side[i]) ROME_HIP(c, hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
    if (!c->ev_side[i]) ROME_HIP(c, hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming));
    if (!c->ev_side2[i]) ROME_HIP(c, hipEventCreateWithFlags(&c->ev_side2[i], hipEventDisableTiming));
  }
  ROME_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  return ROME_OK;
}

int ensure_host(rome_ctx* c, int idx, size_t bytes, double** out) {
  if (bytes == 0) bytes = 8;
  if (c->hcap[idx] < bytes) {
    if (c->hbuf[idx]) { hipError_t e = hipHostFree(c->hbuf[idx]); if (e != hipSuccess) return hip_fail(c, e); c->hbuf[idx] = nullptr; c->hcap[idx] = 0; }
    size_t cap = bytes + bytes / 4;
    hipError_t e = hipHostMalloc(&c->hbuf[idx], cap, hipHostMallocDefault);
    if (e != hipSuccess) return hip_fail(c, e);
    c->hcap[idx] = cap;
  }
  *out = (double*)c->hbuf[idx];
  return ROME_OK;
}

int check_opts(const rome_opts* o) {
  if (!o) return ROME_ERR_INVALID_ARG;
  if (o->n_particles < 1) return ROME_ERR_INVALID_ARG;
  if (o->n_particles > ROME_MAX_PARTICLES) return ROME_ERR_UNSUPPORTED_N;
  if (o->solver < ROME_SOLVER_CLOSED_FORM || o->solver > ROME_SOLVER_GAUSS_NEWTON) return ROME_ERR_INVALID_ARG;
  if (o->max_iters < 1 || o->inflate_cycles < 0 || o->inflate_cycles > 255) return ROME_ERR_INVALID_ARG;
  if (!(o->tol >= 0.0) || !(o->inflation >= 0.0) || !(o->spread_nh >= 0.0)) return ROME_ERR_INVALID_ARG;
  if (!(o->nullhypo >= 0.0 && o->nullhypo <= 1.0)) return ROME_ERR_INVALID_ARG;
  if (o->layout != ROME_LAYOUT_SOA && o->layout != ROME_LAYOUT_AOS && o->layout != ROME_LAYOUT_AOS_POINTS) return ROME_ERR_INVALID_ARG;
  if (o->presampled != ROME_NOISE_STANDARD_NORMALS && o->presampled != ROME_NOISE_MEASUREMENTS) return ROME_ERR_INVALID_ARG;
  return ROME_OK;
}

void fill_args(rome::ConvArgs& a, const rome_opts* o) {
  std::memset(&a, 0, sizeof(a));
  a.N = o->n_particles;
  a.max_iters = o->max_iters;
  a.cycles = o->inflate_cycles;
  a.tol = o->tol;
  a.inflation = o->inflation;
  a.inv_n = 1.0 / (double)o->n_particles;
  a.inv_nm1 = o->n_particles > 1 ? 1.0 / (double)(o->n_particles - 1) : 1.0;
  a.seed = o->seed;
  a.stream_offset = o->stream_offset;
  a.spread_nh = o->spread_nh;
  a.noise_is_meas = o->presampled == ROME_NOISE_MEASUREMENTS;
}

void args_from_dev(rome::ConvArgs& a, const rome_opts* o, const rome_conv_dev* t) {
  fill_args(a, o);
  a.n_conv = t->n_conv; a.dir_all = t->dir_all;
  a.factor = t->factor; a.dir = t->dir; a.fixed_var = t->fixed_var; a.target_var = t->target_var;
  a.mu = t->mu; a.L = t->L; a.bel_fixed = t->bel_fixed; a.bel_target = t->bel_target;
  a.noise = t->noise; a.out = t->out; a.status = t->status;
  a.rows4 = t->rows4;
  a.n_mirror = t->mirror_out ? (t->n_mirror < 0 ? 0 : t->n_mirror) : 0;   // (> 4 is rejected by dev_common)
  for (int m = 0; m < 4; ++m) a.mirror_row[m] = t->mirror_row[m];
  a.mirror_out = t->mirror_out;
  a.mirror_map = t->mirror_out ? t->mirror_map : nullptr;
  a.alt_var = t->hypo_w ? t->alt_var : nullptr;
  a.hypo_w = t->hypo_w;
  a.nullhypo = t->nullhypo;
}

int cholesky_one(int d, const double* cov, double* Lp) {
  double L[36];
  std::memset(L, 0, sizeof(L));
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = cov[i * d + j];
      for (int k = 0; k < j; ++k) s -= L[i * d + k] * L[j * d + k];
      if (i == j) { if (!(s > 0.0)) return ROME_ERR_NOT_POSDEF; L[i * d + i] = std::sqrt(s); }
      else L[i * d + j] = s / L[j * d + j];
    }
  int k = 0;
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) Lp[k++] = L[i * d + j];
  return ROME_OK;
}

// host blocks [C][N][d] (AoS) or [C][d][N] (SoA)  ->  SoA staging (pinned)
void to_soa(const double* src, int C, int N, int d, int layout, double* dst) {
  if (layout == ROME_LAYOUT_SOA) { std::memcpy(dst, src, (size_t)C * N * d * sizeof(double)); return; }
  for (int c = 0; c < C; ++c) {
    const double* s = src + (size_t)c * N * d; double* o = dst + (size_t)c * N * d;
    for (int i = 0; i < N; ++i) for (int k = 0; k < d; ++k) o[(size_t)k * N + i] = s[(size_t)i * d + k];
  }
}
void from_soa(const double* src, int C, int N, int d, int layout, double* dst) {
  if (layout == ROME_LAYOUT_SOA) { std::memcpy(dst, src, (size_t)C * N * d * sizeof(double)); return; }
  for (int c = 0; c < C; ++c) {
    const double* s = src + (size_t)c * N * d; double* o = dst + (size_t)c * N * d;
    for (int i = 0; i < N; ++i) for (int k = 0; k < d; ++k) o[(size_t)i * d + k] = s[(size_t)k * N + i];
  }
}

enum FactorKind { kP2P2, kBR, kP3P3, kPrior2, kPrior3, kPriorPt2 };

inline int point_len(int dim) { return dim == 3 ? 6 : (dim == 6 ? 12 : dim); }

// rows of native points -> rows of coordinates (and back) on the device; host pointers
int convert_rows(rome_ctx* c, int dim, size_t n, const double* src, double* dst, bool to_coords) {
  if (n == 0) return ROME_OK;
  const int pl = point_len(dim);
  if (dim == 2) { std::memcpy(dst, src, sizeof(double) * n * 2); return ROME_OK; }
  ROME_HIP(c, hipSetDevice(c->device));
  void *d_in, *d_out; int rc;
  const size_t nin = sizeof(double) * n * (to_coords ? pl : dim), nout = sizeof(double) * n * (to_coords ? dim : pl);
  if ((rc = ensure(c, 3, nin, &d_in))) return rc;
  if ((rc = ensure(c, 4, nout, &d_out))) return rc;
  ROME_HIP(c, hipMemcpyAsync(d_in, src, nin, hipMemcpyHostToDevice, c->stream));
  ROME_HIP(c, to_coords ? rome::launch_points_to_coords((int)n, dim, (const double*)d_in, (double*)d_out, c->stream)
                        : rome::launch_coords_to_points((int)n, dim, (const double*)d_in, (double*)d_out, c->stream));
  ROME_HIP(c, hipMemcpyAsync(dst, d_out, nout, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}

// common host-pointer path: stage -> launch -> fetch
int host_conv(rome_ctx* ctx, const rome_opts* o, FactorKind kind, int C, const int32_t* dir, int dir_all,
              int dz, int df, int dt, const double* mu, const double* Ltab /*[C][nL]*/, int nL,
              const double* fixed, const double* noise, double* target_inout, int32_t* status,
              const double* alt = nullptr /*C blocks of the other landmark candidate*/, const double* hypo_w = nullptr) {
  const int N = o->n_particles;
  if (o->layout == ROME_LAYOUT_AOS_POINTS) {
    // the reference's native point containers: convert to AoS coordinates on the device, run, convert back
    rome_opts oc = *o; oc.layout = ROME_LAYOUT_AOS;
    const bool has_fx = (kind != kPrior2 && kind != kPrior3 && kind != kPriorPt2);
    const size_t rows = (size_t)C * N;
    std::vector<double> cf, ct(rows * dt), ca;
    int rc2;
    if (has_fx) {
      cf.resize(rows * df);
      if ((rc2 = convert_rows(ctx, df, rows, fixed, cf.data(), true))) return rc2;
      if ((rc2 = convert_rows(ctx, dt, rows, target_inout, ct.data(), true))) return rc2;
    }
    if (alt) { const int dl = dir_all == 1 ? df : dt; ca.resize(rows * dl); if ((rc2 = convert_rows(ctx, dl, rows, alt, ca.data(), true))) return rc2; }
    rc2 = host_conv(ctx, &oc, kind, C, dir, dir_all, dz, df, dt, mu, Ltab, nL, has_fx ? cf.data() : nullptr, noise, ct.data(), status,
                    alt ? ca.data() : nullptr, hypo_w);
    if (rc2) return rc2;
    return convert_rows(ctx, dt, rows, ct.data(), target_inout, false);
  }
  ROME_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const bool has_fixed = (kind != kPrior2 && kind != kPrior3 && kind != kPriorPt2);
  const bool mh = alt && hypo_w;   // multihypo: the alternative landmark blocks are appended behind the landmark-side array
  const size_t blk_f = (size_t)C * N * df, blk_t = (size_t)C * N * dt;
  const size_t n_fixed = has_fixed ? blk_f * ((mh && dir_all == 1) ? 2 : 1) : 0;
  const size_t n_target = has_fixed ? blk_t * ((mh && dir_all != 1) ? 2 : 1) : 0;
  const size_t n_noise = noise ? (size_t)C * N * dz : 0;
  double *h_fixed = nullptr, *h_target = nullptr, *h_noise = nullptr, *h_out = nullptr;
  int rc;
  if (has_fixed) {
    if ((rc = ensure_host(ctx, 0, sizeof(double) * n_fixed, &h_fixed))) return rc;
    if ((rc = ensure_host(ctx, 1, sizeof(double) * n_target, &h_target))) return rc;
    to_soa(fixed, C, N, df, o->layout, h_fixed); to_soa(target_inout, C, N, dt, o->layout, h_target);
  }
  if (noise) { if ((rc = ensure_host(ctx, 2, sizeof(double) * n_noise, &h_noise))) return rc; to_soa(noise, C, N, dz, o->layout, h_noise); }
  if ((rc = ensure_host(ctx, 3, sizeof(double) * blk_t, &h_out))) return rc;
  std::vector<int32_t> h_alt;
  if (mh) {
    if (dir_all == 1) to_soa(alt, C, N, df, o->layout, h_fixed + blk_f); else to_soa(alt, C, N, dt, o->layout, h_target + blk_t);
    h_alt.resize(C);
    for (int c = 0; c < C; ++c) h_alt[c] = C + c;
  }

  void *d_mu, *d_L, *d_fixed = nullptr, *d_target = nullptr, *d_noise = nullptr, *d_out, *d_dir = nullptr, *d_status = nullptr;
  if ((rc = ensure(ctx, 0, sizeof(double) * C * dz, &d_mu))) return rc;
  if ((rc = ensure(ctx, 1, sizeof(double) * C * nL, &d_L))) return rc;
  if ((rc = ensure(ctx, 2, sizeof(double) * (size_t)C * N * dt, &d_out))) return rc;
  ROME_HIP(ctx, hipMemcpyAsync(d_mu, mu, sizeof(double) * C * dz, hipMemcpyHostToDevice, s));
  ROME_HIP(ctx, hipMemcpyAsync(d_L, Ltab, sizeof(double) * C * nL, hipMemcpyHostToDevice, s));
  if (has_fixed) {
    if ((rc = ensure(ctx, 3, sizeof(double) * n_fixed, &d_fixed))) return rc;
    if ((rc = ensure(ctx, 4, sizeof(double) * n_target, &d_target))) return rc;
    ROME_HIP(ctx, hipMemcpyAsync(d_fixed, h_fixed, sizeof(double) * n_fixed, hipMemcpyHostToDevice, s));
    ROME_HIP(ctx, hipMemcpyAsync(d_target, h_target, sizeof(double) * n_target, hipMemcpyHostToDevice, s));
  }
  if (noise) {
    if ((rc = ensure(ctx, 5, sizeof(double) * n_noise, &d_noise))) return rc;
    ROME_HIP(ctx, hipMemcpyAsync(d_noise, h_noise, sizeof(double) * n_noise, hipMemcpyHostToDevice, s));
  }
  if (dir) {
    if ((rc = ensure(ctx, 6, sizeof(int32_t) * C, &d_dir))) return rc;
    ROME_HIP(ctx, hipMemcpyAsync(d_dir, dir, sizeof(int32_t) * C, hipMemcpyHostToDevice, s));
  }
  if (status) { if ((rc = ensure(ctx, 7, sizeof(int32_t) * (size_t)C * N, &d_status))) return rc; }

  void *d_alt = nullptr, *d_hw = nullptr;
  if (!h_alt.empty()) {
    if ((rc = ensure(ctx, 8, sizeof(int32_t) * C, &d_alt))) return rc;
    if ((rc = ensure(ctx, 9, sizeof(double) * C, &d_hw))) return rc;
    ROME_HIP(ctx, hipMemcpyAsync(d_alt, h_alt.data(), sizeof(int32_t) * C, hipMemcpyHostToDevice, s));
    ROME_HIP(ctx, hipMemcpyAsync(d_hw, hypo_w, sizeof(double) * C, hipMemcpyHostToDevice, s));
  }
  void* d_nh = nullptr;
  if (o->nullhypo > 0.0 && has_fixed) {
    std::vector<double> h_nh((size_t)C, o->nullhypo);
    if ((rc = ensure(ctx, 10, sizeof(double) * C, &d_nh))) return rc;
    ROME_HIP(ctx, hipMemcpyAsync(d_nh, h_nh.data(), sizeof(double) * C, hipMemcpyHostToDevice, s));
    ROME_HIP(ctx, hipStreamSynchronize(s));  // h_nh goes out of scope
  }
  rome::ConvArgs a;
  fill_args(a, o);
  a.nullhypo = (const double*)d_nh;
  a.alt_var = (const int32_t*)d_alt; a.hypo_w = (const double*)d_hw;
  a.n_conv = C; a.dir_all = dir_all; a.dir = (const int32_t*)d_dir;
  a.mu = (const double*)d_mu; a.L = (const double*)d_L;
  a.bel_fixed = (const double*)d_fixed; a.bel_target = (const double*)d_target;
  a.noise = (const double*)d_noise; a.out = (double*)d_out; a.status = (int32_t*)d_status;
  hipError_t e = hipSuccess;
  switch (kind) {
    case kP2P2: e = rome::launch_conv_pose2pose2(a, o->solver, s); break;
    case kBR: e = rome::launch_conv_bearingrange(a, o->solver, s); break;
    case kP3P3: e = rome::launch_conv_pose3pose3(a, o->solver, s); break;
    case kPrior2: e = rome::launch_sample_priorpose2(a, s); break;
    case kPrior3: e = rome::launch_sample_priorpose3(a, s); break;
    case kPriorPt2: e = rome::launch_sample_priorpoint2(a, s); break;
  }
  ROME_HIP(ctx, e);
  ROME_HIP(ctx, hipMemcpyAsync(h_out, d_out, sizeof(double) * blk_t, hipMemcpyDeviceToHost, s));
  if (status) ROME_HIP(ctx, hipMemcpyAsync(status, d_status, sizeof(int32_t) * (size_t)C * N, hipMemcpyDeviceToHost, s));
  ROME_HIP(ctx, hipStreamSynchronize(s));
  from_soa(h_out, C, N, dt, o->layout, target_inout);
  return ROME_OK;
}

// rows of doubles: upload inputs, run, download
struct RowBuf { const double* host; int width; };
template <class Launch>
int host_rows(rome_ctx* ctx, int n, const RowBuf* in, int n_in, double* out, int out_width, Launch&& launch) {
  if (!ctx || n < 0 || !out) return ROME_ERR_INVALID_ARG;
  for (int k = 0; k < n_in; ++k) if (!in[k].host && n > 0) return ROME_ERR_INVALID_ARG;
  if (n == 0) return ROME_OK;
  ROME_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  void* d[4] = {nullptr, nullptr, nullptr, nullptr};
  int rc;
  for (int k = 0; k < n_in; ++k) {
    if ((rc = ensure(ctx, k, sizeof(double) * (size_t)n * in[k].width, &d[k]))) return rc;
    ROME_HIP(ctx, hipMemcpyAsync(d[k], in[k].host, sizeof(double) * (size_t)n * in[k].width, hipMemcpyHostToDevice, s));
  }
  void* dout;
  if ((rc = ensure(ctx, 8, sizeof(double) * (size_t)n * out_width, &dout))) return rc;
  ROME_HIP(ctx, launch((const double*)d[0], (const double*)d[1], (const double*)d[2], (double*)dout, s));
  ROME_HIP(ctx, hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n * out_width, hipMemcpyDeviceToHost, s));
  ROME_HIP(ctx, hipStreamSynchronize(s));
  return ROME_OK;
}

}  // namespace

extern "C" {

int rome_version(void) { return ROME_MI355_VERSION; }

const char* rome_strerror(int code) {
  switch (code) {
    case ROME_OK: return "ok";
    case ROME_ERR_INVALID_ARG: return "invalid argument";
    case ROME_ERR_NO_DEVICE: return "no HIP device available (librome_mi355 has no CPU fallback)";
    case ROME_ERR_HIP: return "HIP runtime error (see rome_last_hip_error_string)";
    case ROME_ERR_NOT_POSDEF: return "covariance is not positive definite";
    case ROME_ERR_UNSUPPORTED_N: return "n_particles exceeds ROME_MAX_PARTICLES";
    case ROME_ERR_ALLOC: return "host allocation failed";
    default: return "unknown error";
  }
}
int rome_last_hip_error(const rome_ctx* ctx) { return ctx ? (int)ctx->last_hip : 0; }
const char* rome_last_hip_error_string(const rome_ctx* ctx) { return hipGetErrorString(ctx ? ctx->last_hip : hipSuccess); }

void rome_opts_default(rome_opts* o, int32_t solver) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->n_particles = 100;
  o->solver = solver;
  o->max_iters = solver == ROME_SOLVER_NELDER_MEAD ? 1000 : 20;
  o->inflate_cycles = 3;
  o->tol = solver == ROME_SOLVER_NELDER_MEAD ? 1e-8 : 1e-12;
  o->inflation = 5.0;
  o->seed = 0x524F4D45ull; /* "ROME" */
  o->stream_offset = 0;
  o->layout = ROME_LAYOUT_SOA;
  o->spread_nh = 3.0;
}

int rome_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rome_ctx_create(rome_ctx** out, int device) {
  if (!out) return ROME_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ROME_ERR_NO_DEVICE;
  if (device < 0 || device >= n) return ROME_ERR_INVALID_ARG;
  rome_ctx* c = new (std::nothrow) rome_ctx();
  if (!c) return ROME_ERR_ALLOC;
  c->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return ROME_ERR_HIP; }
  c->stream = c->own_stream;
  *out = c;
  return ROME_OK;
}

void rome_ctx_destroy(rome_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int i = 0; i < rome_ctx::kBufs; ++i) if (c->dbuf[i]) (void)hipFree(c->dbuf[i]);
  for (int i = 0; i < rome_ctx::kHostBufs; ++i) if (c->hbuf[i]) (void)hipHostFree(c->hbuf[i]);
  for (int i = 0; i < rome_ctx::kSide; ++i) {
    if (c->side[i]) { (void)hipStreamSynchronize(c->side[i]); (void)hipStreamDestroy(c->side[i]); }
    if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
    if (c->ev_side2[i]) (void)hipEventDestroy(c->ev_side2[i]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_order) (void)hipEventDestroy(c->ev_order);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int rome_ctx_set_stream(rome_ctx* c, void* hip_stream) {
  if (!c) return ROME_ERR_INVALID_ARG;
  if (c->stream != (hipStream_t)hip_stream) {
    // Everything launched through a context is ONE logical sequence whatever stream it runs on: belief stores, plans' arenas, the
    // context's workspaces and caller tensors written by an earlier entry are read by later ones.  A stream change therefore orders
    // the new stream after ALL work queued on the previous one -- unconditionally, by an event (no host synchronisation, a few µs):
    // the previous stream may be the private non-blocking one, which nothing else ever joins (round 5's race: a block operation on
    // the private stream, then a plan run on the caller's stream reading its blocks).
    ROME_BIND(c);
    if (!c->ev_order) ROME_HIP(c, hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
    ROME_HIP(c, hipEventRecord(c->ev_order, c->stream));
    ROME_HIP(c, hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_order, 0));
  }
  c->stream = (hipStream_t)hip_stream;  // NULL is HIP's default (null) stream, e.g. torch's default stream
  return ROME_OK;
}
int rome_ctx_use_own_stream(rome_ctx* c) {
  if (!c) return ROME_ERR_INVALID_ARG;
  return rome_ctx_set_stream(c, (void*)c->own_stream);
}
int rome_ctx_synchronize(rome_ctx* c) {
  if (!c) return ROME_ERR_INVALID_ARG;
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}

int rome_cholesky_lower(int32_t d, int32_t n, const double* cov, double* L) {
  if (d < 1 || d > 6 || n < 0 || (n > 0 && (!cov || !L))) return ROME_ERR_INVALID_ARG;
  const int nL = d * (d + 1) / 2;
  for (int i = 0; i < n; ++i) {
    int rc = cholesky_one(d, cov + (size_t)i * d * d, L + (size_t)i * nL);
    if (rc) return rc;
  }
  return ROME_OK;
}

/* ---- residual entry points ---- */
int rome_residual_pose2pose2(rome_ctx* c, int32_t n, const double* z, const double* p, const double* q, double* r) {
  RowBuf in[3] = {{z, 3}, {p, 3}, {q, 3}};
  return host_rows(c, n, in, 3, r, 3, [&](const double* a, const double* b, const double* d, double* o, hipStream_t s) {
    return rome::launch_residual_pose2pose2(n, a, b, d, o, s); });
}
int rome_residual_priorpose2(rome_ctx* c, int32_t n, const double* m, const double* p, double* r) {
  RowBuf in[2] = {{m, 3}, {p, 3}};
  return host_rows(c, n, in, 2, r, 3, [&](const double* a, const double* b, const double*, double* o, hipStream_t s) {
    return rome::launch_residual_priorpose2(n, a, b, o, s); });
}
int rome_residual_pose2point2br(rome_ctx* c, int32_t n, const double* z, const double* p, const double* l, double* r) {
  RowBuf in[3] = {{z, 2}, {p, 3}, {l, 2}};
  return host_rows(c, n, in, 3, r, 2, [&](const double* a, const double* b, const double* d, double* o, hipStream_t s) {
    return rome::launch_residual_bearingrange(n, a, b, 0, d, o, s); });
}
int rome_residual_pose2point2br_pt(rome_ctx* c, int32_t n, const double* z, const double* p, const double* l, double* r) {
  RowBuf in[3] = {{z, 2}, {p, 6}, {l, 2}};
  return host_rows(c, n, in, 3, r, 2, [&](const double* a, const double* b, const double* d, double* o, hipStream_t s) {
    return rome::launch_residual_bearingrange(n, a, b, 1, d, o, s); });
}
int rome_residual_pose3pose3(rome_ctx* c, int32_t n, const double* z, const double* p, const double* q, double* r) {
  RowBuf in[3] = {{z, 6}, {p, 6}, {q, 6}};
  return host_rows(c, n, in, 3, r, 6, [&](const double* a, const double* b, const double* d, double* o, hipStream_t s) {
    return rome::launch_residual_pose3pose3(n, a, b, d, 0, o, s); });
}
int rome_residual_pose3pose3_pt(rome_ctx* c, int32_t n, const double* z, const double* p, const double* q, double* r) {
  RowBuf in[3] = {{z, 6}, {p, 12}, {q, 12}};
  return host_rows(c, n, in, 3, r, 6, [&](const double* a, const double* b, const double* d, double* o, hipStream_t s) {
    return rome::launch_residual_pose3pose3(n, a, b, d, 1, o, s); });
}
int rome_residual_priorpose3(rome_ctx* c, int32_t n, const double* m, const double* p, double* r) {
  RowBuf in[2] = {{m, 6}, {p, 6}};
  return host_rows(c, n, in, 2, r, 6, [&](const double* a, const double* b, const double*, double* o, hipStream_t s) {
    return rome::launch_residual_priorpose3(n, a, b, o, s); });
}

/* ---- host-pointer convolutions ---- */
int rome_conv_pose2pose2(rome_ctx* c, const rome_opts* o, int32_t C, const int32_t* dir, const double* mu, const double* cov,
                         const double* fixed, const double* noise, double* target_inout, int32_t* status) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (C > 0 && (!mu || !cov || !fixed || !target_inout))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  std::vector<double> L((size_t)C * 6);
  if ((rc = rome_cholesky_lower(3, C, cov, L.data()))) return rc;
  return host_conv(c, o, kP2P2, C, dir, 0, 3, 3, 3, mu, L.data(), 6, fixed, noise, target_inout, status);
}
int rome_conv_pose2pose2_mh(rome_ctx* c, const rome_opts* o, int32_t C, int32_t dir, const double* mu, const double* cov,
                            const double* fixed, const double* alt, const double* hypo_w, const double* noise,
                            double* target_inout, int32_t* status) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (dir != 0 && dir != 1) || (C > 0 && (!mu || !cov || !fixed || !alt || !hypo_w || !target_inout))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  for (int i = 0; i < C; ++i) if (!(hypo_w[i] >= 0.0 && hypo_w[i] <= 1.0)) return ROME_ERR_INVALID_ARG;
  std::vector<double> L((size_t)C * 6);
  if ((rc = rome_cholesky_lower(3, C, cov, L.data()))) return rc;
  return host_conv(c, o, kP2P2, C, nullptr, dir, 3, 3, 3, mu, L.data(), 6, fixed, noise, target_inout, status, alt, hypo_w);
}
int rome_conv_pose2point2br(rome_ctx* c, const rome_opts* o, int32_t C, int32_t dir, const double* mu, const double* sigma,
                            const double* fixed, const double* noise, double* target_inout, int32_t* status) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (dir != 0 && dir != 1) || (C > 0 && (!mu || !sigma || !fixed || !target_inout))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  for (int i = 0; i < 2 * C; ++i) if (sigma[i] != sigma[i]) return ROME_ERR_NOT_POSDEF;  /* sigma < 0 encodes Uniform(mu ± |sigma|) */
  return host_conv(c, o, kBR, C, nullptr, dir, 2, dir == 0 ? 3 : 2, dir == 0 ? 2 : 3, mu, sigma, 2, fixed, noise, target_inout, status);
}
int rome_conv_pose2point2br_mh(rome_ctx* c, const rome_opts* o, int32_t C, int32_t dir, const double* mu, const double* sigma,
                               const double* fixed, const double* alt, const double* hypo_w, const double* noise,
                               double* target_inout, int32_t* status) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (dir != 0 && dir != 1) || (C > 0 && (!mu || !sigma || !fixed || !alt || !hypo_w || !target_inout))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  for (int i = 0; i < 2 * C; ++i) if (sigma[i] != sigma[i]) return ROME_ERR_NOT_POSDEF;
  for (int i = 0; i < C; ++i) if (!(hypo_w[i] >= 0.0 && hypo_w[i] <= 1.0)) return ROME_ERR_INVALID_ARG;
  return host_conv(c, o, kBR, C, nullptr, dir, 2, dir == 0 ? 3 : 2, dir == 0 ? 2 : 3, mu, sigma, 2, fixed, noise, target_inout, status, alt, hypo_w);
}
int rome_conv_pose3pose3(rome_ctx* c, const rome_opts* o, int32_t C, const int32_t* dir, const double* mu, const double* cov,
                         const double* fixed, const double* noise, double* target_inout, int32_t* status) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (C > 0 && (!mu || !cov || !fixed || !target_inout))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  std::vector<double> L((size_t)C * 21);
  if ((rc = rome_cholesky_lower(6, C, cov, L.data()))) return rc;
  return host_conv(c, o, kP3P3, C, dir, 0, 6, 6, 6, mu, L.data(), 21, fixed, noise, target_inout, status);
}
int rome_sample_priorpose2(rome_ctx* c, const rome_opts* o, int32_t C, const double* mu, const double* cov, const double* noise, double* out) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (C > 0 && (!mu || !cov || !out))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  std::vector<double> L((size_t)C * 6);
  if ((rc = rome_cholesky_lower(3, C, cov, L.data()))) return rc;
  return host_conv(c, o, kPrior2, C, nullptr, 0, 3, 3, 3, mu, L.data(), 6, nullptr, noise, out, nullptr);
}
int rome_sample_priorpose3(rome_ctx* c, const rome_opts* o, int32_t C, const double* mu, const double* cov, const double* noise, double* out) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (C > 0 && (!mu || !cov || !out))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  std::vector<double> L((size_t)C * 21);
  if ((rc = rome_cholesky_lower(6, C, cov, L.data()))) return rc;
  return host_conv(c, o, kPrior3, C, nullptr, 0, 6, 6, 6, mu, L.data(), 21, nullptr, noise, out, nullptr);
}

int rome_sample_priorpoint2(rome_ctx* c, const rome_opts* o, int32_t C, const double* mu, const double* cov, const double* noise, double* out) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || C < 0 || (C > 0 && (!mu || !cov || !out))) return ROME_ERR_INVALID_ARG;
  if (C == 0) return ROME_OK;
  std::vector<double> L((size_t)C * 3);
  if ((rc = rome_cholesky_lower(2, C, cov, L.data()))) return rc;
  rome_opts oc = *o;
  if (oc.layout == ROME_LAYOUT_AOS_POINTS) oc.layout = ROME_LAYOUT_AOS;   // a Point2 point IS its coordinates
  return host_conv(c, &oc, kPriorPt2, C, nullptr, 0, 2, 2, 2, mu, L.data(), 3, nullptr, noise, out, nullptr);
}

/* ---- device-pointer convolutions ---- */
static int dev_common(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t, bool need_beliefs) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || !t || t->n_conv < 0) return ROME_ERR_INVALID_ARG;
  if (t->n_conv > 0 && (!t->mu || !t->L || !t->out)) return ROME_ERR_INVALID_ARG;
  if (t->n_conv > 0 && need_beliefs && (!t->bel_fixed || !t->bel_target)) return ROME_ERR_INVALID_ARG;
  if (t->mirror_out && !t->mirror_map && t->n_mirror > 4) return ROME_ERR_INVALID_ARG;   // mirror_row holds 4 rows (never silently dropped); more: mirror_map
  ROME_BIND(c);
  return ROME_OK;
}
int rome_conv_pose2pose2_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t) {
  int rc = dev_common(c, o, t, true); if (rc) return rc;
  rome::ConvArgs a; args_from_dev(a, o, t);
  ROME_HIP(c, rome::launch_conv_pose2pose2(a, o->solver, c->stream));
  return ROME_OK;
}
int rome_conv_pose2point2br_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t) {
  int rc = dev_common(c, o, t, true); if (rc) return rc;
  if (t->dir != nullptr || (t->dir_all != 0 && t->dir_all != 1)) return ROME_ERR_INVALID_ARG;
  rome::ConvArgs a; args_from_dev(a, o, t);
  ROME_HIP(c, rome::launch_conv_bearingrange(a, o->solver, c->stream));
  return ROME_OK;
}
int rome_sweep_pose2_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* p2p2, const rome_conv_dev* br1, const rome_conv_dev* br0,
                         const uint64_t* family_stream_offset) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || (!p2p2 && !br1 && !br0)) return ROME_ERR_INVALID_ARG;
  const rome_conv_dev* t[3] = {p2p2, br1, br0};
  rome::ConvArgs a[3];
  for (int k = 0; k < 3; ++k) {
    if (!t[k]) continue;
    if ((rc = dev_common(c, o, t[k], true))) return rc;
    if (k > 0 && (t[k]->dir != nullptr || t[k]->dir_all != (k == 1 ? 1 : 0))) return ROME_ERR_INVALID_ARG;
    rome_opts of = *o;
    if (family_stream_offset) of.stream_offset = o->stream_offset + family_stream_offset[k];
    args_from_dev(a[k], &of, t[k]);
  }
  ROME_HIP(c, rome::launch_sweep_pose2(p2p2 ? &a[0] : nullptr, br1 ? &a[1] : nullptr, br0 ? &a[2] : nullptr, o->solver, c->stream));
  return ROME_OK;
}
int rome_conv_pose3pose3_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t) {
  int rc = dev_common(c, o, t, true); if (rc) return rc;
  rome::ConvArgs a; args_from_dev(a, o, t);
  ROME_HIP(c, rome::launch_conv_pose3pose3(a, o->solver, c->stream));
  return ROME_OK;
}
int rome_sample_priorpose2_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t) {
  int rc = dev_common(c, o, t, false); if (rc) return rc;
  rome::ConvArgs a; args_from_dev(a, o, t);
  ROME_HIP(c, rome::launch_sample_priorpose2(a, c->stream));
  return ROME_OK;
}
int rome_sample_priorpose3_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t) {
  int rc = dev_common(c, o, t, false); if (rc) return rc;
  rome::ConvArgs a; args_from_dev(a, o, t);
  ROME_HIP(c, rome::launch_sample_priorpose3(a, c->stream));
  return ROME_OK;
}
int rome_sample_priorpoint2_dev(rome_ctx* c, const rome_opts* o, const rome_conv_dev* t) {
  int rc = dev_common(c, o, t, false); if (rc) return rc;
  rome::ConvArgs a; args_from_dev(a, o, t);
  ROME_HIP(c, rome::launch_sample_priorpoint2(a, c->stream));
  return ROME_OK;
}

/* ---- clique-level batch from host beliefs ---- */
namespace {
// one belief array of the clique: [n][dim][N] (SoA), [n][N][dim] (AoS) or native points -> SoA coordinates on the device
int stage_beliefs(rome_ctx* c, const rome_opts* o, int n, int dim, const double* host, std::vector<double>& tmp, void** dev, size_t* used,
                  unsigned char* arena, size_t cap) {
  const int N = o->n_particles;
  const size_t cnt = (size_t)n * dim * N;
  *dev = arena + *used;
  if (n == 0) return ROME_OK;
  if (*used + cnt * sizeof(double) > cap) return ROME_ERR_ALLOC;
  const double* src = host;
  if (o->layout == ROME_LAYOUT_AOS_POINTS && dim != 2) {
    tmp.resize(cnt);   // points -> AoS coordinates (device conversion kernels), then the transpose below
    int rc = convert_rows(c, dim, (size_t)n * N, host, tmp.data(), true); if (rc) return rc;
    std::vector<double> soa(cnt);
    to_soa(tmp.data(), n, N, dim, ROME_LAYOUT_AOS, soa.data());
    tmp.swap(soa); src = tmp.data();
  } else if (o->layout != ROME_LAYOUT_SOA) {
    tmp.resize(cnt); to_soa(host, n, N, dim, ROME_LAYOUT_AOS, tmp.data()); src = tmp.data();
  }
  ROME_HIP(c, hipMemcpyAsync(*dev, src, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (src != host) ROME_HIP(c, hipStreamSynchronize(c->stream));   // tmp is reused by the caller
  *used += (cnt * sizeof(double) + 255) & ~(size_t)255;
  return ROME_OK;
}
// SoA device blocks -> host blocks in the caller's layout (synchronises)
int fetch_beliefs(rome_ctx* c, int layout, int n, int dim, int N, const double* dev, double* host) {
  if (n == 0) return ROME_OK;
  const size_t cnt = (size_t)n * dim * N;
  if (layout == ROME_LAYOUT_SOA) {
    ROME_HIP(c, hipMemcpyAsync(host, dev, cnt * 8, hipMemcpyDeviceToHost, c->stream));
    ROME_HIP(c, hipStreamSynchronize(c->stream));
    return ROME_OK;
  }
  std::vector<double> soa(cnt);
  ROME_HIP(c, hipMemcpyAsync(soa.data(), dev, cnt * 8, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  if (layout == ROME_LAYOUT_AOS || dim == 2) { from_soa(soa.data(), n, N, dim, ROME_LAYOUT_AOS, host); return ROME_OK; }
  std::vector<double> aos(cnt);
  from_soa(soa.data(), n, N, dim, ROME_LAYOUT_AOS, aos.data());
  return convert_rows(c, dim, (size_t)n * N, aos.data(), host, false);
}
// host vectors feed / receive asynchronous copies: whatever way a clique entry returns (an error in the middle included), the stream is
// drained before those vectors are destroyed (declare the guard AFTER them)
struct DrainOnExit { hipStream_t s; ~DrainOnExit() { (void)hipStreamSynchronize(s); } };

// the five row families of a rome_clique_host.  kind: 0 Pose2Pose2 (+ PriorPose2 rows), 1 bearing-range, 2 Pose3Pose3 (+ PriorPose3 rows),
// 3 PriorPoint2 sampler; variable types: 0 Pose2, 1 Point2, 2 Pose3; valt = type of the array an `alt` entry indexes (-1: no multihypo);
// base = first row of the family in the proposal buffer of its target type (rome_clique_upsolve)
struct Fam {
  int n; const int32_t* rows4; int F; const double* mu; const double* spread; int dz, nL, dfx, dt; double* out; int vf, vt; int dir_all;
  uint64_t off; int kind; int base;
  const int32_t* alt; const double* hw; const double* nh; const int32_t* sid; int valt;
  const int32_t* meas;   // per-row block of the measurement samples (Pose2Pose2 rows only), or nullptr
};
constexpr int NF = 5;
void make_fams(const rome_clique_host* q, Fam (&fam)[NF]) {
  const Fam f[NF] = {
    {q->n_p2p2, q->p2p2_rows4, q->f_p2p2, q->p2p2_mu, q->p2p2_cov, 3, 6, 3, 3, q->out_p2p2, 0, 0, 0, 0ull, 0, 0,
     q->p2p2_alt, q->p2p2_hypo_w, q->p2p2_nullhypo, q->p2p2_stream, 0, q->p2p2_meas},
    {q->n_br1, q->br1_rows4, q->f_br, q->br_mu, q->br_sigma, 2, 2, 2, 3, q->out_br1, 1, 0, 1, 1ull << 28, 1, q->n_p2p2,
     q->br1_alt, q->br1_hypo_w, q->br1_nullhypo, q->br1_stream, 1, q->br1_meas},
    {q->n_br0, q->br0_rows4, q->f_br, q->br_mu, q->br_sigma, 2, 2, 3, 2, q->out_br0, 0, 1, 0, 2ull << 28, 1, 0,
     q->br0_alt, q->br0_hypo_w, q->br0_nullhypo, q->br0_stream, 1, q->br0_meas},
    {q->n_p3p3, q->p3p3_rows4, q->f_p3p3, q->p3p3_mu, q->p3p3_cov, 6, 21, 6, 6, q->out_p3p3, 2, 2, 0, 5ull << 28, 2, 0,
     nullptr, nullptr, q->p3p3_nullhypo, q->p3p3_stream, -1, nullptr},
    {q->n_prpt2, q->prpt2_rows4, q->f_prpt2, q->prpt2_mu, q->prpt2_cov, 2, 3, 2, 2, q->out_prpt2, 1, 1, 0, 7ull << 28, 3, q->n_br0,
     nullptr, nullptr, nullptr, q->prpt2_stream, -1, nullptr}};
  for (int k = 0; k < NF; ++k) fam[k] = f[k];
}
// table entries must address the arrays they index (nv = variables per type); hypothesis / stream columns in range
int check_fam_rows(const Fam& f, const int (&nv)[3]) {
  if (f.n < 0 || f.F < 0 || (f.n > 0 && (!f.rows4 || !f.mu || !f.spread || f.F == 0))) return ROME_ERR_INVALID_ARG;
  if (f.alt && !f.hw) return ROME_ERR_INVALID_ARG;
  for (int r = 0; r < f.n; ++r) {
    const int32_t* e = f.rows4 + 4 * (size_t)r;
    if (e[0] < 0 || e[0] >= f.F || e[2] < 0 || e[2] >= nv[f.vf] || e[3] < 0 || e[3] >= nv[f.vt] || e[1] < 0 || e[1] > 2) return ROME_ERR_INVALID_ARG;
    if (f.alt && f.alt[r] >= 0) {
      if (f.alt[r] >= nv[f.valt] || !(f.hw[r] >= 0.0 && f.hw[r] <= 1.0)) return ROME_ERR_INVALID_ARG;
    } else if (f.alt && f.alt[r] < -1) return ROME_ERR_INVALID_ARG;
    if (f.nh && !(f.nh[r] >= 0.0 && f.nh[r] <= 1.0)) return ROME_ERR_INVALID_ARG;
    if (f.sid && (f.sid[r] < 0 || f.sid[r] >= (1 << 28))) return ROME_ERR_INVALID_ARG;
    if (f.meas && (f.meas[r] < -1 || f.meas[r] >= nv[f.kind == 1 ? 1 : 0] || (f.meas[r] >= 0 && e[1] == 2))) return ROME_ERR_INVALID_ARG;
  }
  return ROME_OK;
}
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
// bytes of a family's device tables (rows4, mu, L, the optional columns)
size_t fam_table_bytes(const Fam& f) {
  return al256((size_t)f.n * 16) + al256((size_t)f.F * f.dz * 8) + al256((size_t)f.F * f.nL * 8) + 256 +
         (f.alt ? al256((size_t)f.n * 4) + al256((size_t)f.n * 8) : 0) + (f.nh ? al256((size_t)f.n * 8) : 0) + (f.sid ? al256((size_t)f.n * 4) : 0) +
         (f.meas ? al256((size_t)f.n * 4) : 0);
}
struct FamDev { const int32_t* rows = nullptr; const double* mu = nullptr; const double* L = nullptr; const int32_t* alt = nullptr;
                const double* hw = nullptr; const double* nh = nullptr; const int32_t* sid = nullptr; const int32_t* meas = nullptr; };
// uploads a family's tables into `arena` (asynchronously: `Ls` must outlive the stream work); MvNormal factors get their packed Cholesky
int upload_fam(rome_ctx* c, const Fam& f, unsigned char* arena, size_t* used, std::vector<double>& Ls, FamDev& d) {
  d = FamDev{};
  if (f.n == 0) return ROME_OK;
  hipStream_t s = c->stream;
  const double* Lsrc = f.spread;
  int rc;
  if (f.kind != 1) {   // MvNormal factors: packed lower Cholesky of every covariance
    Ls.resize((size_t)f.F * f.nL);
    if ((rc = rome_cholesky_lower(f.dz, f.F, f.spread, Ls.data()))) return rc;
    Lsrc = Ls.data();
  }
  auto put = [&](const void* src, size_t bytes) -> void* {
    void* dst = arena + *used;
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
    *used += al256(bytes);
    return dst;
  };
#define ROME_PUT(dst, T, src, bytes) do { void* _p = put((src), (bytes)); if (!_p) return hip_fail(c, hipGetLastError()); (dst) = (const T*)_p; } while (0)
  ROME_PUT(d.rows, int32_t, f.rows4, (size_t)f.n * 16);
  ROME_PUT(d.mu, double, f.mu, (size_t)f.F * f.dz * 8);
  ROME_PUT(d.L, double, Lsrc, (size_t)f.F * f.nL * 8);
  if (f.alt) { ROME_PUT(d.alt, int32_t, f.alt, (size_t)f.n * 4); ROME_PUT(d.hw, double, f.hw, (size_t)f.n * 8); }
  if (f.nh) ROME_PUT(d.nh, double, f.nh, (size_t)f.n * 8);
  if (f.sid) ROME_PUT(d.sid, int32_t, f.sid, (size_t)f.n * 4);
  if (f.meas) ROME_PUT(d.meas, int32_t, f.meas, (size_t)f.n * 4);
#undef ROME_PUT
  return ROME_OK;
}
// rows [lo, hi) of a family: one convolution launch into `out` (the block of row `lo`)
hipError_t launch_fam(const Fam& f, const FamDev& d, const rome_opts* o, uint64_t stream_base, int lo, int hi, const double* bel_fixed,
                      const double* bel_target, double* out, hipStream_t s, const double* meas_base = nullptr) {
  rome::ConvArgs a;
  rome_opts of = *o;
  of.stream_offset = stream_base + f.off + (d.sid ? 0ull : (uint64_t)lo);   // family offsets of the device graph (DeviceGraph.STREAM_*)
  fill_args(a, &of);
  a.n_conv = hi - lo; a.dir_all = f.dir_all; a.rows4 = d.rows + 4 * (size_t)lo;
  a.mu = d.mu; a.L = d.L;
  a.bel_fixed = bel_fixed; a.bel_target = bel_target; a.out = out;
  a.alt_var = d.alt ? d.alt + lo : nullptr; a.hypo_w = d.alt ? d.hw + lo : nullptr;
  a.nullhypo = d.nh ? d.nh + lo : nullptr;
  a.row_stream = d.sid ? d.sid + lo : nullptr;
  if (d.meas && meas_base) { a.meas_block = d.meas + lo; a.meas_base = meas_base; }
  return f.kind == 0 ? rome::launch_conv_pose2pose2(a, o->solver, s) : (f.kind == 1 ? rome::launch_conv_bearingrange(a, o->solver, s)
                     : (f.kind == 2 ? rome::launch_conv_pose3pose3(a, o->solver, s) : rome::launch_sample_priorpoint2(a, s)));
}
}  // namespace

int rome_clique_proposals(rome_ctx* c, const rome_opts* o, const rome_clique_host* q) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || !q) return ROME_ERR_INVALID_ARG;
  const int N = o->n_particles;
  Fam fam[NF]; make_fams(q, fam);
  const int nv[3] = {q->n_pose2, q->n_point2, q->n_pose3};
  const int vdim[3] = {3, 2, 6};
  const double* vhost[3] = {q->bel_pose2, q->bel_point2, q->bel_pose3};
  size_t need = 0;
  for (int t = 0; t < 3; ++t) { if (nv[t] < 0 || (nv[t] > 0 && !vhost[t])) return ROME_ERR_INVALID_ARG; need += al256((size_t)nv[t] * vdim[t] * N * 8); }
  for (const Fam& f : fam) {
    if ((rc = check_fam_rows(f, nv))) return rc;
    if (f.n > 0 && !f.out) return ROME_ERR_INVALID_ARG;
    if (f.meas) return ROME_ERR_INVALID_ARG;   // measurement-sample rows address a store (rome_upsolve_plan)
    need += al256((size_t)f.n * f.dt * N * 8) + fam_table_bytes(f);
  }
  ROME_BIND(c);
  void* arena_v = nullptr;
  if ((rc = ensure(c, 9, need + 4096, &arena_v))) return rc;
  unsigned char* arena = (unsigned char*)arena_v;
  size_t used = 0;
  const size_t cap = need + 4096;
  std::vector<double> tmp;
  void* dbel[3];
  for (int t = 0; t < 3; ++t) if ((rc = stage_beliefs(c, o, nv[t], vdim[t], vhost[t], tmp, &dbel[t], &used, arena, cap))) return rc;
  hipStream_t s = c->stream;
  std::vector<std::vector<double>> Ls(NF), hout(NF);
  DrainOnExit drain{s};
  double* dout[NF] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < NF; ++k) {
    const Fam& f = fam[k];
    if (f.n == 0) continue;
    FamDev d;
    if ((rc = upload_fam(c, f, arena, &used, Ls[k], d))) return rc;
    dout[k] = (double*)(arena + used);
    used += al256((size_t)f.n * f.dt * N * 8);
    ROME_HIP(c, launch_fam(f, d, o, o->stream_offset, 0, f.n, (const double*)dbel[f.vf], (const double*)dbel[f.vt], dout[k], s));
  }
  // proposals back to the host in the caller's layout
  for (int k = 0; k < NF; ++k) {
    const Fam& f = fam[k];
    if (f.n == 0) continue;
    const size_t cnt = (size_t)f.n * f.dt * N;
    if (o->layout == ROME_LAYOUT_SOA) ROME_HIP(c, hipMemcpyAsync(f.out, dout[k], cnt * 8, hipMemcpyDeviceToHost, s));
    else { hout[k].resize(cnt); ROME_HIP(c, hipMemcpyAsync(hout[k].data(), dout[k], cnt * 8, hipMemcpyDeviceToHost, s)); }
  }
  ROME_HIP(c, hipStreamSynchronize(s));
  if (o->layout != ROME_LAYOUT_SOA)
    for (int k = 0; k < NF; ++k) {
      const Fam& f = fam[k];
      if (f.n == 0) continue;
      if (o->layout == ROME_LAYOUT_AOS || f.dt == 2) from_soa(hout[k].data(), f.n, N, f.dt, ROME_LAYOUT_AOS, f.out);
      else {
        std::vector<double> aos(hout[k].size());
        from_soa(hout[k].data(), f.n, N, f.dt, ROME_LAYOUT_AOS, aos.data());
        if ((rc = convert_rows(c, f.dt, (size_t)f.n * N, aos.data(), f.out, false))) return rc;
      }
    }
  return ROME_OK;
}

}  // extern "C"

/* ---- belief store + up-solve plans: gibbs_iters x {proposals -> manikde! bandwidths -> multiscale Gibbs product -> in-place write},
 *      device-resident across calls ---- */
struct rome_store {
  rome_ctx* ctx = nullptr;
  int N = 0;
  int nv[3] = {0, 0, 0};
  double* bel[3] = {nullptr, nullptr, nullptr};
  bool owned = false;
};
struct rome_scatter_plan {
  rome_ctx* ctx = nullptr; rome_store* st = nullptr;
  int n = 0; int64_t stride = 0;
  int32_t* d_ent = nullptr;   // [n][4] = (dim, var, src_block, type)
};
struct rome_blockop_plan {
  rome_ctx* ctx = nullptr; rome_store* st = nullptr;
  int op = 0, n = 0;
  int32_t* d_ent = nullptr;   // [n][4] = (type, a, b, dst)
  double* d_prm = nullptr;    // COMPOSE: [n][2] = (translation, heading) inflation of the composed deviations, or NULL
};
struct rome_upsolve_plan {
  rome_ctx* ctx = nullptr; rome_store* st = nullptr;
  int N = 0, layout = 0, gi = 3, pi = 1, n_up = 0;
  Fam fam[NF]; FamDev fd[NF];
  std::vector<int> fam_lo[NF];            // first row of family f that targets update position >= k (rows are grouped in update order)
  std::vector<int> step_k;                // boundaries of the update steps (ranges of the update list updated together)
  std::vector<int> up_cnt_before;         // [k][t]: updates of type t among the first k
  int n_upt[3] = {0, 0, 0}, prop_rows_t[3] = {0, 0, 0}, max_k[3] = {1, 1, 1};
  double* d_prop[3] = {nullptr, nullptr, nullptr}; double* d_pbw[3] = {nullptr, nullptr, nullptr};
  int32_t* d_ptr[3] = {nullptr, nullptr, nullptr}; int32_t* d_rws[3] = {nullptr, nullptr, nullptr};
  int32_t* d_upblock[3] = {nullptr, nullptr, nullptr}; int32_t* d_upstream[3] = {nullptr, nullptr, nullptr};
  int32_t* d_upmirror[3] = {nullptr, nullptr, nullptr};
  int32_t* d_gather[3] = {nullptr, nullptr, nullptr};   // (dim, var, position, type) per updated variable: store -> contiguous download buffer
  double* d_newout[3] = {nullptr, nullptr, nullptr}; double* d_bwout[3] = {nullptr, nullptr, nullptr};
  double* new_host[3] = {nullptr, nullptr, nullptr}; double* bw_host[3] = {nullptr, nullptr, nullptr};
  bool has_mirror = false, has_upstream = false;
  int n_smsg[3] = {0, 0, 0}, smsg_base[3] = {0, 0, 0};   // store-resident messages: rows [smsg_base, smsg_base + n_smsg) of the type's proposal buffer
  int32_t* d_smsg_ent[3] = {nullptr, nullptr, nullptr};  // (dim, source block, proposal row, type) per message: gathered at the start of every run
  void* arena = nullptr; bool arena_owned = false;
  size_t tree_need = 8;               // the tree workspaces of the three types side by side (their products may run concurrently)
  size_t tree_off[3] = {0, 0, 0};
};

namespace {
const int kVdim[3] = {3, 2, 6};
const uint32_t kCircBw[3] = {0b100u, 0u, 0b111000u}, kCircProd[3] = {0b100u, 0u, 0u};
const uint64_t kProdOff[3] = {3ull << 28, 4ull << 28, 6ull << 28};

// builds a plan over `st`; ctx_arena: carve the plan's device memory from the context's arena (the one-shot host entry) instead of an
// allocation the plan owns
int plan_build(rome_ctx* c, rome_store* st, const rome_opts* o, const rome_clique_upsolve_host* u, rome_upsolve_plan* P, bool ctx_arena) {
  const rome_clique_host* q = &u->clique;
  const int N = o->n_particles;
  if (N > ROME_MAX_PARTICLES_GIBBS) return ROME_ERR_UNSUPPORTED_N;   // the multiscale Gibbs product: lane = output sample, two wavefronts per variable
  if (N != st->N) return ROME_ERR_INVALID_ARG;
  if (N < 2 || u->n_up < 0 || (u->n_up > 0 && (!u->up_type || !u->up_var))) return ROME_ERR_INVALID_ARG;
  if (u->schedule != ROME_UPSOLVE_SEQUENTIAL && u->schedule != ROME_UPSOLVE_JACOBI) return ROME_ERR_INVALID_ARG;
  P->ctx = c; P->st = st; P->N = N; P->layout = o->layout; P->n_up = u->n_up;
  P->gi = u->gibbs_iters > 0 ? u->gibbs_iters : 3; P->pi = u->product_iters > 0 ? u->product_iters : 1;
  const int nv[3] = {st->nv[0], st->nv[1], st->nv[2]};
  if (q->n_pose2 > nv[0] || q->n_point2 > nv[1] || q->n_pose3 > nv[2]) return ROME_ERR_INVALID_ARG;
  // ---- update list: (type, variable) -> global position k and position within the type's list
  std::vector<int> kpos[3], uplist[3];
  for (int t = 0; t < 3; ++t) kpos[t].assign((size_t)nv[t], -1);
  P->up_cnt_before.assign((size_t)u->n_up * 3 + 3, 0);
  for (int k = 0; k < u->n_up; ++k) {
    const int t = u->up_type[k], v = u->up_var[k];
    if (t < 0 || t > 2 || v < 0 || v >= nv[t] || kpos[t][v] >= 0) return ROME_ERR_INVALID_ARG;
    if (u->up_stream && (u->up_stream[k] < 0 || u->up_stream[k] >= (1 << 28))) return ROME_ERR_INVALID_ARG;
    if (u->up_mirror && u->up_mirror[k] < -1) return ROME_ERR_INVALID_ARG;
    kpos[t][v] = k;
    for (int tt = 0; tt < 3; ++tt) P->up_cnt_before[3 * (size_t)(k + 1) + tt] = P->up_cnt_before[3 * (size_t)k + tt] + (tt == t ? 1 : 0);
    uplist[t].push_back(k);
  }
  make_fams(q, P->fam);
  const int n_msg[3] = {u->n_msg_pose2, u->n_msg_point2, u->n_msg_pose3};
  const double* msg_host[3] = {u->msg_pose2, u->msg_point2, u->msg_pose3};
  const int32_t* msg_up[3] = {u->msg_pose2_up, u->msg_point2_up, u->msg_pose3_up};
  double* new_host[3] = {u->new_pose2, u->new_point2, u->new_pose3};
  double* bw_host[3] = {u->bw_pose2, u->bw_point2, u->bw_pose3};
  const int n_smsg[3] = {u->n_smsg_pose2, u->n_smsg_point2, u->n_smsg_pose3};
  const int32_t* smsg_src[3] = {u->smsg_pose2_src, u->smsg_point2_src, u->smsg_pose3_src};
  const int32_t* smsg_up[3] = {u->smsg_pose2_up, u->smsg_point2_up, u->smsg_pose3_up};
  int msg_base[3];
  int rc;
  for (int t = 0; t < 3; ++t) P->prop_rows_t[t] = 0;
  for (const Fam& f : P->fam) { if ((rc = check_fam_rows(f, nv))) return rc; P->prop_rows_t[f.vt] += f.n; }
  for (int t = 0; t < 3; ++t) {
    if (n_msg[t] < 0 || (n_msg[t] > 0 && (!msg_host[t] || !msg_up[t]))) return ROME_ERR_INVALID_ARG;
    P->n_upt[t] = (int)uplist[t].size();
    P->new_host[t] = P->n_upt[t] ? new_host[t] : nullptr; P->bw_host[t] = P->n_upt[t] ? bw_host[t] : nullptr;
    msg_base[t] = P->prop_rows_t[t]; P->prop_rows_t[t] += n_msg[t];
    if (n_smsg[t] < 0 || (n_smsg[t] > 0 && (!smsg_src[t] || !smsg_up[t]))) return ROME_ERR_INVALID_ARG;
    P->n_smsg[t] = n_smsg[t]; P->smsg_base[t] = P->prop_rows_t[t]; P->prop_rows_t[t] += n_smsg[t];
  }
  P->has_mirror = u->up_mirror != nullptr; P->has_upstream = u->up_stream != nullptr;
  // ---- rows: every row targets an updated variable, rows grouped in update order; the row range of every update position; CSR
  std::vector<std::vector<int>> csr[3];    // per type: proposal rows (in the type's buffer) of every updated variable, in update order
  for (int t = 0; t < 3; ++t) csr[t].resize(uplist[t].size());
  for (int k4 = 0; k4 < NF; ++k4) {
    const Fam& f = P->fam[k4];
    P->fam_lo[k4].assign((size_t)u->n_up + 1, 0);
    int prev = -1;
    for (int r = 0; r < f.n; ++r) {
      const int32_t* e = f.rows4 + 4 * (size_t)r;
      const int k = kpos[f.vt][e[3]];
      if (k < 0 || k < prev) return ROME_ERR_INVALID_ARG;
      if (k != prev) { for (int kk = prev + 1; kk <= k; ++kk) P->fam_lo[k4][kk] = r; }
      prev = k;
      csr[f.vt][(size_t)(P->up_cnt_before[3 * (size_t)k + f.vt])].push_back(f.base + r);
    }
    for (int kk = prev + 1; kk <= u->n_up; ++kk) P->fam_lo[k4][kk] = f.n;
  }
  for (int t = 0; t < 3; ++t)
    for (int m = 0; m < n_msg[t]; ++m) {
      const int k = msg_up[t][m];
      if (k < 0 || k >= u->n_up || u->up_type[k] != t) return ROME_ERR_INVALID_ARG;
      csr[t][(size_t)P->up_cnt_before[3 * (size_t)k + t]].push_back(msg_base[t] + m);
    }
  std::vector<int32_t> sm_h[3];
  for (int t = 0; t < 3; ++t)
    for (int m = 0; m < n_smsg[t]; ++m) {   // store-resident messages: source = a block of the store that this plan does not write
      const int k = smsg_up[t][m], src = smsg_src[t][m];
      if (k < 0 || k >= u->n_up || u->up_type[k] != t || src < 0 || src >= nv[t] || kpos[t][src] >= 0) return ROME_ERR_INVALID_ARG;
      csr[t][(size_t)P->up_cnt_before[3 * (size_t)k + t]].push_back(P->smsg_base[t] + m);
      sm_h[t].push_back(kVdim[t]); sm_h[t].push_back(src); sm_h[t].push_back(P->smsg_base[t] + m); sm_h[t].push_back(t);
    }
  std::vector<int32_t> ptr_h[3], rws_h[3], blk_h[3], sid_h[3], mir_h[3], gat_h[3];
  for (int t = 0; t < 3; ++t) {
    P->max_k[t] = 1;
    ptr_h[t].push_back(0);
    for (const auto& l : csr[t]) { for (int r : l) rws_h[t].push_back(r); ptr_h[t].push_back((int32_t)rws_h[t].size()); if ((int)l.size() > P->max_k[t]) P->max_k[t] = (int)l.size(); }
    if (rws_h[t].empty()) rws_h[t].push_back(0);
    int pos = 0;
    for (int k : uplist[t]) {
      blk_h[t].push_back(u->up_var[k]);
      sid_h[t].push_back(u->up_stream ? u->up_stream[k] : pos);
      mir_h[t].push_back(u->up_mirror ? u->up_mirror[k] : -1);
      gat_h[t].push_back(kVdim[t]); gat_h[t].push_back(u->up_var[k]); gat_h[t].push_back(pos); gat_h[t].push_back(t);
      ++pos;
    }
  }
  // ---- steps: ranges [k0, k1) of the update list that are updated together
  P->step_k.clear(); P->step_k.push_back(0);
  if (u->up_group) {
    for (int k = 1; k < u->n_up; ++k) {
      if (u->up_group[k] < u->up_group[k - 1]) return ROME_ERR_INVALID_ARG;
      if (u->up_group[k] != u->up_group[k - 1]) P->step_k.push_back(k);
    }
  } else if (u->schedule == ROME_UPSOLVE_SEQUENTIAL) {
    for (int k = 1; k < u->n_up; ++k) P->step_k.push_back(k);
  }
  P->step_k.push_back(u->n_up);
  // ---- device memory
  size_t need = 4096;
  for (int t = 0; t < 3; ++t) {
    const size_t blk = (size_t)kVdim[t] * N * 8, nu = uplist[t].size();
    need += al256((size_t)P->prop_rows_t[t] * blk) + al256((size_t)P->prop_rows_t[t] * kVdim[t] * 8) + al256(ptr_h[t].size() * 4) + al256(rws_h[t].size() * 4)
          + 3 * al256(nu * 4 + 4) + al256(nu * 16 + 16) + al256(nu * blk) + al256(nu * kVdim[t] * 8) + al256((size_t)n_smsg[t] * 16 + 16);
  }
  for (const Fam& f : P->fam) need += fam_table_bytes(f);
  ROME_BIND(c);
  if (ctx_arena) { if ((rc = ensure(c, 12, need, &P->arena))) return rc; P->arena_owned = false; }
  else { ROME_HIP(c, hipMalloc(&P->arena, need)); P->arena_owned = true; }
  unsigned char* arena = (unsigned char*)P->arena;
  size_t used = 0;
  hipStream_t s = c->stream;
  std::vector<std::vector<double>> Ls(NF);
  std::vector<double> tmp;
  DrainOnExit drain{s};
  auto put = [&](const void* src, size_t bytes, void** dst) -> int {
    *dst = arena + used;
    if (bytes) ROME_HIP(c, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s));
    used += al256(bytes);
    return ROME_OK;
  };
  auto take = [&](size_t bytes) -> void* { void* p = arena + used; used += al256(bytes); return p; };
  for (int t = 0; t < 3; ++t) {
    const size_t blk = (size_t)kVdim[t] * N * 8, nu = uplist[t].size();
    P->d_prop[t] = (double*)take((size_t)P->prop_rows_t[t] * blk);
    P->d_pbw[t] = (double*)take((size_t)P->prop_rows_t[t] * kVdim[t] * 8);
    P->d_newout[t] = (double*)take(nu * blk);
    P->d_bwout[t] = (double*)take(nu * kVdim[t] * 8);
    void* p;
    if ((rc = put(ptr_h[t].data(), ptr_h[t].size() * 4, &p))) return rc; P->d_ptr[t] = (int32_t*)p;
    if ((rc = put(rws_h[t].data(), rws_h[t].size() * 4, &p))) return rc; P->d_rws[t] = (int32_t*)p;
    if ((rc = put(blk_h[t].data(), nu * 4, &p))) return rc; P->d_upblock[t] = (int32_t*)p;
    if ((rc = put(sid_h[t].data(), nu * 4, &p))) return rc; P->d_upstream[t] = (int32_t*)p;
    if ((rc = put(mir_h[t].data(), nu * 4, &p))) return rc; P->d_upmirror[t] = (int32_t*)p;
    if ((rc = put(gat_h[t].data(), nu * 16, &p))) return rc; P->d_gather[t] = (int32_t*)p;
    if ((rc = put(sm_h[t].data(), (size_t)n_smsg[t] * 16, &p))) return rc; P->d_smsg_ent[t] = (int32_t*)p;
    if (n_msg[t] > 0) {   // upward messages: appended to the type's proposal buffer, bandwidths once
      void* dm = nullptr; size_t used_m = 0;
      unsigned char* mb = (unsigned char*)(P->d_prop[t] + (size_t)msg_base[t] * kVdim[t] * N);
      if ((rc = stage_beliefs(c, o, n_msg[t], kVdim[t], msg_host[t], tmp, &dm, &used_m, mb, (size_t)n_msg[t] * blk + 256))) return rc;
      ROME_HIP(c, rome::launch_kde_bandwidth(kVdim[t], n_msg[t], N, (const double*)mb, kCircBw[t], 1e-2, 1e-6,
                                             P->d_pbw[t] + (size_t)msg_base[t] * kVdim[t], nullptr, s));
    }
    P->tree_off[t] = t == 0 ? 0 : P->tree_need;
    if (t == 0) P->tree_need = 0;
    P->tree_need += al256(rome::gibbs_workspace_bytes(kVdim[t], P->prop_rows_t[t], (int)nu, N)) + 256;
  }
  for (int k4 = 0; k4 < NF; ++k4) if ((rc = upload_fam(c, P->fam[k4], arena, &used, Ls[k4], P->fd[k4]))) return rc;
  if (used > need) return ROME_ERR_ALLOC;
  // the host tables must not be referenced after creation (the caller's arrays may go away)
  for (Fam& f : P->fam) { f.rows4 = nullptr; f.mu = f.spread = nullptr; f.alt = nullptr; f.hw = f.nh = nullptr; f.sid = nullptr; f.out = nullptr; f.meas = nullptr; }
  return ROME_OK;   // (~DrainOnExit: the uploads have completed before Ls / tmp go away)
}

int plan_run(rome_upsolve_plan* P, const rome_opts* o, double* mirror_out, int64_t mirror_stride) {
  rome_ctx* c = P->ctx; rome_store* st = P->st;
  const int N = P->N;
  int rc;
  if (P->has_mirror && !mirror_out) return ROME_ERR_INVALID_ARG;
  if (mirror_stride == 0) mirror_stride = 6 * (int64_t)N;
  if (P->has_mirror)
    if (mirror_stride < (int64_t)N) return ROME_ERR_INVALID_ARG;   // (a block spans dim * N doubles from its slot: PACKED layouts use stride N)
  ROME_BIND(c);
  void* trees = nullptr;
  if ((rc = ensure(c, 10, P->tree_need, &trees))) return rc;
  hipStream_t s = c->stream;
  // A step is two phases of mutually independent launch chains: (A) per row family {convolutions -> manikde! bandwidths of those
  // proposals} -- they read the store, write disjoint proposal rows --, then (B) per variable type {ball trees -> multiscale Gibbs
  // product} -- each writes its own type's blocks in place.  All of (A) precedes all of (B), and all of (B) the next step's (A): a
  // product of one type overwrites beliefs that another family's convolution reads.  A phase with more than one chain runs its
  // chains on side streams of the context (a small clique or frontier pays the LATENCY of its launches: the landmark product need
  // not wait for the pose product; 1.1 -> 0.6 ms per Gibbs iteration on the 36-pose honeycomb, profiles/r04_small_frontier.txt),
  // and consecutive phases are chained DIRECTLY by events -- every chain of a phase waits for the events of the previous phase's
  // chains, one cross-stream hop, not a join into the context's stream followed by a fork out of it (each hop is 20-40 us of queue
  // latency on this stack); the context's stream is joined once at the end.  A phase with a single chain after work that is
  // already on the context's stream stays there: a Manhattan frontier (one family, one type) never leaves the one stream.
  static const bool no_fork = std::getenv("ROME_UPSOLVE_NO_FORK") != nullptr;   // (A/B measurements: everything on the one stream)
  hipEvent_t* ev_cur = c->ev_side;      // events of the phase whose completion the next phase waits for (when !on_main)
  hipEvent_t* ev_nxt = c->ev_side2;
  int n_cur = 0;
  bool on_main = true;                  // everything issued so far is ordered on the context's stream itself
  auto phase = [&](int n_chain, auto&& launch_chain) -> int {
    if (n_chain == 0) return ROME_OK;
    const bool side = n_chain > 1 && !no_fork;
    int rc2;
    if (side || !on_main) { if ((rc2 = ensure_side(c))) return rc2; }
    if (side && on_main) ROME_HIP(c, hipEventRecord(c->ev_fork, s));
    for (int i = 0; i < n_chain; ++i) {
      hipStream_t sx = side ? c->side[i] : s;
      if (on_main) { if (side) ROME_HIP(c, hipStreamWaitEvent(sx, c->ev_fork, 0)); }
      else for (int e = 0; e < n_cur; ++e) ROME_HIP(c, hipStreamWaitEvent(sx, ev_cur[e], 0));
      if ((rc2 = launch_chain(i, sx))) return rc2;
      if (side) ROME_HIP(c, hipEventRecord(ev_nxt[i], sx));
    }
    if (side) { hipEvent_t* t_ = ev_cur; ev_cur = ev_nxt; ev_nxt = t_; n_cur = n_chain; on_main = false; }
    else on_main = true;
    return ROME_OK;
  };
  // store-resident messages: the source blocks as they are NOW -> their proposal rows, and their manikde! bandwidths (once per run)
  for (int t = 0; t < 3; ++t)
    if (P->n_smsg[t] > 0 && P->gi > 0) {
      ROME_HIP(c, rome::launch_scatter_blocks(P->n_smsg[t], N, P->d_smsg_ent[t], P->d_prop[t], (int64_t)kVdim[t] * N, st->bel[0], st->bel[1], st->bel[2], s, /*to_store=*/0));
      ROME_HIP(c, rome::launch_kde_bandwidth(kVdim[t], P->n_smsg[t], N, P->d_prop[t] + (size_t)P->smsg_base[t] * kVdim[t] * N, kCircBw[t], 1e-2, 1e-6,
                                             P->d_pbw[t] + (size_t)P->smsg_base[t] * kVdim[t], nullptr, s));
    }
  for (int it = 0; it < P->gi; ++it) {
    const uint64_t base = o->stream_offset + ((uint64_t)it << 32);
    const int nsteps = P->n_up > 0 ? (int)P->step_k.size() - 1 : 0;
    for (int stp = 0; stp < nsteps; ++stp) {
      const int k0 = P->step_k[stp], k1 = P->step_k[stp + 1];
      int fam_a[NF], lo_a[NF], hi_a[NF], naf = 0;
      for (int k4 = 0; k4 < NF; ++k4) {
        const Fam& f = P->fam[k4];
        const int lo = P->fam_lo[k4][k0], hi = f.n == 0 ? 0 : (k1 < P->n_up ? P->fam_lo[k4][k1] : f.n);
        if (hi > lo) { fam_a[naf] = k4; lo_a[naf] = lo; hi_a[naf] = hi; ++naf; }
      }
      int typ_a[3], nat = 0;
      for (int t = 0; t < 3; ++t) {
        const int pa = P->up_cnt_before[3 * (size_t)k0 + t], pb = P->up_cnt_before[3 * (size_t)k1 + t];
        if (pb > pa && P->prop_rows_t[t] > 0) typ_a[nat++] = t;
      }
      rc = phase(naf, [&](int i, hipStream_t sx) -> int {
        const int k4 = fam_a[i], lo = lo_a[i], hi = hi_a[i];
        const Fam& f = P->fam[k4];
        double* out = P->d_prop[f.vt] + (size_t)(f.base + lo) * f.dt * N;
        ROME_HIP(c, launch_fam(f, P->fd[k4], o, base, lo, hi, st->bel[f.vf], st->bel[f.vt], out, sx, st->bel[f.kind == 1 ? 1 : 0]));
        if (P->max_k[f.vt] > 1)   // (a plan whose destinations take ONE proposal each -- sampling a graph's measurements, transporting a belief -- multiplies nothing: no manikde!)
          ROME_HIP(c, rome::launch_kde_bandwidth(f.dt, hi - lo, N, out, kCircBw[f.vt], 1e-2, 1e-6, P->d_pbw[f.vt] + (size_t)(f.base + lo) * f.dt, nullptr, sx));
        return ROME_OK;
      });
      if (rc) return rc;
      rc = phase(nat, [&](int i, hipStream_t sx) -> int {
        const int t = typ_a[i];
        const int pa = P->up_cnt_before[3 * (size_t)k0 + t], pb = P->up_cnt_before[3 * (size_t)k1 + t];
        // the product writes the new beliefs IN PLACE into the store (a product reads only proposals and its own variable's block)
        rome::GibbsPlace place{P->d_upblock[t] + pa, P->has_upstream ? P->d_upstream[t] + pa : nullptr,
                               (P->has_mirror && mirror_out) ? P->d_upmirror[t] + pa : nullptr, mirror_out, mirror_stride};
        ROME_HIP(c, rome::launch_product_gibbs(kVdim[t], pb - pa, N, P->prop_rows_t[t], P->d_ptr[t] + pa, P->d_rws[t], P->d_prop[t], P->d_pbw[t],
                                               st->bel[t], st->bel[t], (unsigned char*)trees + P->tree_off[t], kCircProd[t], P->pi, P->max_k[t],
                                               o->seed, base + kProdOff[t] + (P->has_upstream ? 0ull : (uint64_t)pa), sx, &place));
        return ROME_OK;
      });
      if (rc) return rc;
    }
  }
  if (!on_main) {   // the one join: everything after the run is ordered after it on the context's stream
    for (int e = 0; e < n_cur; ++e) ROME_HIP(c, hipStreamWaitEvent(s, ev_cur[e], 0));
    on_main = true;
  }
  if (P->has_mirror && P->gi > 0) {
    // updated variables whose product never ran (no proposals at all) still owe their block to the mirror: the product kernel handles
    // K = 0 (copy), so nothing to do here as long as the type has proposal rows; a type without any row keeps its beliefs
    for (int t = 0; t < 3; ++t)
      if (P->n_upt[t] && P->prop_rows_t[t] == 0) {
        rome::GibbsPlace place{P->d_upblock[t], nullptr, P->d_upmirror[t], mirror_out, mirror_stride};
        ROME_HIP(c, rome::launch_product_gibbs(kVdim[t], P->n_upt[t], N, 0, P->d_ptr[t], P->d_rws[t], P->d_prop[t], P->d_pbw[t], st->bel[t], st->bel[t],
                                               (unsigned char*)trees + P->tree_off[t], kCircProd[t], 1, 1, o->seed, 0, s, &place));
      }
  }
  // ---- results (only when the plan was created with host outputs): the updated beliefs and their manikde! bandwidths
  bool sync = false;
  for (int t = 0; t < 3; ++t) {
    const int nu = P->n_upt[t];
    if (nu == 0) continue;
    if (P->bw_host[t]) {
      ROME_HIP(c, rome::launch_kde_bandwidth(kVdim[t], nu, N, st->bel[t], kCircBw[t], 1e-2, 1e-6, P->d_bwout[t], nullptr, s, P->d_upblock[t]));
      ROME_HIP(c, hipMemcpyAsync(P->bw_host[t], P->d_bwout[t], (size_t)nu * kVdim[t] * 8, hipMemcpyDeviceToHost, s));
      sync = true;
    }
    if (P->new_host[t]) {
      ROME_HIP(c, rome::launch_scatter_blocks(nu, N, P->d_gather[t], P->d_newout[t], (int64_t)kVdim[t] * N, st->bel[0], st->bel[1], st->bel[2], s, /*to_store=*/0));
      if ((rc = fetch_beliefs(c, P->layout, nu, kVdim[t], N, P->d_newout[t], P->new_host[t]))) return rc;
    }
  }
  if (sync) ROME_HIP(c, hipStreamSynchronize(s));
  return ROME_OK;
}
}  // namespace

extern "C" {

int rome_store_create(rome_ctx* c, int32_t N, int32_t n_pose2, int32_t n_point2, int32_t n_pose3, rome_store** out) {
  if (!c || !out || N < 1 || N > ROME_MAX_PARTICLES || n_pose2 < 0 || n_point2 < 0 || n_pose3 < 0) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  rome_store* st = new (std::nothrow) rome_store();
  if (!st) return ROME_ERR_ALLOC;
  st->ctx = c; st->N = N; st->nv[0] = n_pose2; st->nv[1] = n_point2; st->nv[2] = n_pose3; st->owned = true;
  for (int t = 0; t < 3; ++t) {
    const size_t bytes = (size_t)st->nv[t] * kVdim[t] * N * 8;
    hipError_t e = hipMalloc((void**)&st->bel[t], bytes ? bytes : 8);
    if (e == hipSuccess && bytes) e = hipMemsetAsync(st->bel[t], 0, bytes, c->stream);
    if (e != hipSuccess) { for (int k = 0; k <= t; ++k) if (st->bel[k]) (void)hipFree(st->bel[k]); delete st; return hip_fail(c, e); }
  }
  *out = st;
  return ROME_OK;
}
int rome_store_wrap(rome_ctx* c, int32_t N, int32_t n_pose2, double* d2, int32_t n_point2, double* dpt, int32_t n_pose3, double* d3, rome_store** out) {
  if (!c || !out || N < 1 || N > ROME_MAX_PARTICLES || n_pose2 < 0 || n_point2 < 0 || n_pose3 < 0) return ROME_ERR_INVALID_ARG;
  if ((n_pose2 > 0 && !d2) || (n_point2 > 0 && !dpt) || (n_pose3 > 0 && !d3)) return ROME_ERR_INVALID_ARG;
  rome_store* st = new (std::nothrow) rome_store();
  if (!st) return ROME_ERR_ALLOC;
  st->ctx = c; st->N = N; st->nv[0] = n_pose2; st->nv[1] = n_point2; st->nv[2] = n_pose3; st->owned = false;
  st->bel[0] = d2; st->bel[1] = dpt; st->bel[2] = d3;
  *out = st;
  return ROME_OK;
}
void rome_store_destroy(rome_store* st) {
  if (!st) return;
  if (st->owned) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != st->ctx->device) (void)hipSetDevice(st->ctx->device);
    for (int t = 0; t < 3; ++t) if (st->bel[t]) (void)hipFree(st->bel[t]);
  }
  delete st;
}
int rome_store_upload(rome_store* st, int32_t layout, int32_t type, int32_t first, int32_t count, const double* host) {
  if (!st || type < 0 || type > 2 || first < 0 || count < 0 || (int64_t)first + count > st->nv[type] || (count > 0 && !host)) return ROME_ERR_INVALID_ARG;
  if (layout != ROME_LAYOUT_SOA && layout != ROME_LAYOUT_AOS && layout != ROME_LAYOUT_AOS_POINTS) return ROME_ERR_INVALID_ARG;
  if (count == 0) return ROME_OK;
  rome_ctx* c = st->ctx;
  ROME_BIND(c);
  rome_opts o; rome_opts_default(&o, ROME_SOLVER_NEWTON); o.n_particles = st->N; o.layout = layout;
  std::vector<double> tmp;
  DrainOnExit drain{c->stream};
  void* dv = nullptr; size_t used = 0;
  const size_t blk = (size_t)kVdim[type] * st->N;
  int rc = stage_beliefs(c, &o, count, kVdim[type], host, tmp, &dv, &used, (unsigned char*)(st->bel[type] + (size_t)first * blk), (size_t)count * blk * 8 + 256);
  if (rc) return rc;
  ROME_HIP(c, hipStreamSynchronize(c->stream));   // the caller's array may go away
  return ROME_OK;
}
int rome_store_download(rome_store* st, int32_t layout, int32_t type, int32_t first, int32_t count, double* host) {
  if (!st || type < 0 || type > 2 || first < 0 || count < 0 || (int64_t)first + count > st->nv[type] || (count > 0 && !host)) return ROME_ERR_INVALID_ARG;
  if (layout != ROME_LAYOUT_SOA && layout != ROME_LAYOUT_AOS && layout != ROME_LAYOUT_AOS_POINTS) return ROME_ERR_INVALID_ARG;
  rome_ctx* c = st->ctx;
  ROME_BIND(c);
  return fetch_beliefs(c, layout, count, kVdim[type], st->N, st->bel[type] + (size_t)first * kVdim[type] * st->N, host);
}
int rome_store_ptr(rome_store* st, int32_t type, void** dev, int32_t* n_blocks) {
  if (!st || type < 0 || type > 2 || !dev) return ROME_ERR_INVALID_ARG;
  *dev = st->bel[type];
  if (n_blocks) *n_blocks = st->nv[type];
  return ROME_OK;
}

int rome_upsolve_plan_create(rome_ctx* c, rome_store* st, const rome_opts* o, const rome_clique_upsolve_host* u, rome_upsolve_plan** out) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || !st || !u || !out || st->ctx != c) return ROME_ERR_INVALID_ARG;
  rome_upsolve_plan* P = new (std::nothrow) rome_upsolve_plan();
  if (!P) return ROME_ERR_ALLOC;
  rc = plan_build(c, st, o, u, P, false);
  if (rc) { rome_upsolve_plan_destroy(P); return rc; }
  *out = P;
  return ROME_OK;
}
int rome_upsolve_plan_run(rome_upsolve_plan* P, const rome_opts* o, double* mirror_out, int64_t mirror_stride) {
  int rc = check_opts(o); if (rc) return rc;
  if (!P || o->n_particles != P->N || mirror_stride < 0) return ROME_ERR_INVALID_ARG;
  return plan_run(P, o, mirror_out, mirror_stride);
}
void rome_upsolve_plan_destroy(rome_upsolve_plan* P) {
  if (!P) return;
  if (P->arena && P->arena_owned) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != P->ctx->device) (void)hipSetDevice(P->ctx->device);
    (void)hipFree(P->arena);
  }
  delete P;
}

int rome_scatter_plan_create(rome_ctx* c, rome_store* st, int32_t n, const int32_t* type, const int32_t* var, const int32_t* src_block,
                             int64_t stride, rome_scatter_plan** out) {
  if (!c || !st || st->ctx != c || !out || n < 0 || stride < 0 || (n > 0 && (!type || !var || !src_block))) return ROME_ERR_INVALID_ARG;
  if (stride == 0) stride = 6 * (int64_t)st->N;
  std::vector<int32_t> ent((size_t)n * 4 + 4);
  for (int k = 0; k < n; ++k) {
    const int t = type[k];
    if (t < 0 || t > 2 || var[k] < 0 || var[k] >= st->nv[t] || src_block[k] < 0 || stride < (int64_t)st->N) return ROME_ERR_INVALID_ARG;
    ent[4 * (size_t)k] = kVdim[t]; ent[4 * (size_t)k + 1] = var[k]; ent[4 * (size_t)k + 2] = src_block[k]; ent[4 * (size_t)k + 3] = t;
  }
  ROME_BIND(c);
  rome_scatter_plan* S = new (std::nothrow) rome_scatter_plan();
  if (!S) return ROME_ERR_ALLOC;
  S->ctx = c; S->st = st; S->n = n; S->stride = stride;
  hipError_t e = hipMalloc((void**)&S->d_ent, (size_t)n * 16 + 16);
  if (e == hipSuccess && n) e = hipMemcpy(S->d_ent, ent.data(), (size_t)n * 16, hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (S->d_ent) (void)hipFree(S->d_ent); delete S; return hip_fail(c, e); }
  *out = S;
  return ROME_OK;
}
int rome_blockop_plan_create(rome_ctx* c, rome_store* st, int32_t op, int32_t n, const int32_t* type, const int32_t* a, const int32_t* b,
                             const int32_t* dst, rome_blockop_plan** out) {
  return rome_blockop_plan_create_ex(c, st, op, n, type, a, b, dst, nullptr, out);
}
int rome_blockop_plan_create_ex(rome_ctx* c, rome_store* st, int32_t op, int32_t n, const int32_t* type, const int32_t* a, const int32_t* b,
                                const int32_t* dst, const double* params, rome_blockop_plan** out) {
  if (params && op != ROME_BLOCKOP_COMPOSE) return ROME_ERR_INVALID_ARG;
  if (params) for (int k = 0; k < 2 * n; ++k) if (!(params[k] > 0.0) || !(params[k] < 1e6)) return ROME_ERR_INVALID_ARG;
  if (!c || !st || !out || st->ctx != c || n < 0 || op < ROME_BLOCKOP_COPY || op > ROME_BLOCKOP_MIX) return ROME_ERR_INVALID_ARG;
  if (n > 0 && (!type || !a || !dst || ((op == ROME_BLOCKOP_RELATIVE || op == ROME_BLOCKOP_COMPOSE) && !b))) return ROME_ERR_INVALID_ARG;
  std::vector<int32_t> ent((size_t)n * 4 + 4, 0);
  for (int k = 0; k < n; ++k) {
    const int t = type[k] & 0xff, fl = type[k] >> 8;
    if (type[k] < 0 || t > 2 || a[k] < 0 || a[k] >= st->nv[op == ROME_BLOCKOP_RELATIVE ? 0 : t] || dst[k] < 0 || dst[k] >= st->nv[t]) return ROME_ERR_INVALID_ARG;
    if (op != ROME_BLOCKOP_COMPOSE && op != ROME_BLOCKOP_MIX && fl != 0) return ROME_ERR_INVALID_ARG;
    if (op == ROME_BLOCKOP_MIX && (fl < 1 || dst[k] == a[k])) return ROME_ERR_INVALID_ARG;
    if (op == ROME_BLOCKOP_RELATIVE && (t > 1 || b[k] < 0 || b[k] >= st->nv[t] || a[k] >= st->nv[0])) return ROME_ERR_INVALID_ARG;
    if (op == ROME_BLOCKOP_COMPOSE && (t != 0 || fl > 3 || b[k] < 0 || b[k] >= st->nv[0] || dst[k] == a[k] || dst[k] == b[k])) return ROME_ERR_INVALID_ARG;
    ent[4 * (size_t)k] = type[k]; ent[4 * (size_t)k + 1] = a[k]; ent[4 * (size_t)k + 2] = b ? b[k] : 0; ent[4 * (size_t)k + 3] = dst[k];
  }
  ROME_BIND(c);
  rome_blockop_plan* B = new (std::nothrow) rome_blockop_plan();
  if (!B) return ROME_ERR_ALLOC;
  B->ctx = c; B->st = st; B->op = op; B->n = n;
  if (hipMalloc((void**)&B->d_ent, (size_t)n * 16 + 16) != hipSuccess) { delete B; return hip_fail(c, hipGetLastError()); }
  if (hipMemcpy(B->d_ent, ent.data(), (size_t)n * 16 + 16, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(B->d_ent); delete B; return hip_fail(c, hipGetLastError()); }
  if (params && n > 0) {
    if (hipMalloc((void**)&B->d_prm, (size_t)n * 16) != hipSuccess || hipMemcpy(B->d_prm, params, (size_t)n * 16, hipMemcpyHostToDevice) != hipSuccess) {
      if (B->d_prm) (void)hipFree(B->d_prm);
      (void)hipFree(B->d_ent); delete B; return hip_fail(c, hipGetLastError());
    }
  }
  *out = B;
  return ROME_OK;
}
int rome_blockop_plan_run(rome_blockop_plan* B) {
  if (!B) return ROME_ERR_INVALID_ARG;
  rome_ctx* c = B->ctx;
  ROME_BIND(c);
  ROME_HIP(c, rome::launch_block_ops(B->op, B->n, B->st->N, B->d_ent, B->st->bel[0], B->st->bel[1], B->st->bel[2], c->stream, B->d_prm));
  return ROME_OK;
}
void rome_blockop_plan_destroy(rome_blockop_plan* B) {
  if (!B) return;
  if (B->d_ent) (void)hipFree(B->d_ent);
  if (B->d_prm) (void)hipFree(B->d_prm);
  delete B;
}

int rome_scatter_plan_run(rome_scatter_plan* S, const double* src_dev) {
  if (!S || (S->n > 0 && !src_dev)) return ROME_ERR_INVALID_ARG;
  rome_ctx* c = S->ctx;
  ROME_BIND(c);
  ROME_HIP(c, rome::launch_scatter_blocks(S->n, S->st->N, S->d_ent, src_dev, S->stride, S->st->bel[0], S->st->bel[1], S->st->bel[2], c->stream, 1));
  return ROME_OK;
}
void rome_scatter_plan_destroy(rome_scatter_plan* S) {
  if (!S) return;
  if (S->d_ent) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != S->ctx->device) (void)hipSetDevice(S->ctx->device);
    (void)hipFree(S->d_ent);
  }
  delete S;
}

/* ---- the one-shot host entry: a temporary store over the clique's host beliefs (context arena) + a plan + one run ---- */
int rome_clique_upsolve(rome_ctx* c, const rome_opts* o, const rome_clique_upsolve_host* u) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || !u) return ROME_ERR_INVALID_ARG;
  const rome_clique_host* q = &u->clique;
  const int N = o->n_particles;
  if (N > ROME_MAX_PARTICLES_GIBBS) return ROME_ERR_UNSUPPORTED_N;
  const int nv[3] = {q->n_pose2, q->n_point2, q->n_pose3};
  const double* vhost[3] = {q->bel_pose2, q->bel_point2, q->bel_pose3};
  size_t need = 4096;
  for (int t = 0; t < 3; ++t) { if (nv[t] < 0 || (nv[t] > 0 && !vhost[t])) return ROME_ERR_INVALID_ARG; need += al256((size_t)nv[t] * kVdim[t] * N * 8); }
  if (u->up_mirror) return ROME_ERR_INVALID_ARG;   // (mirrors belong to plans: there is no device buffer to mirror into here)
  for (int k = 0; k < u->n_up; ++k) {               // the results are read back: every updated type needs its host outputs
    const int t = u->up_type ? u->up_type[k] : -1;
    if (t < 0 || t > 2) return ROME_ERR_INVALID_ARG;
    double* nh[3] = {u->new_pose2, u->new_point2, u->new_pose3}; double* bh[3] = {u->bw_pose2, u->bw_point2, u->bw_pose3};
    if (!nh[t] || !bh[t]) return ROME_ERR_INVALID_ARG;
  }
  ROME_BIND(c);
  void* arena_v = nullptr;
  if ((rc = ensure(c, 11, need, &arena_v))) return rc;
  unsigned char* arena = (unsigned char*)arena_v;
  size_t used = 0;
  rome_store st;
  st.ctx = c; st.N = N; st.owned = false;
  {
    std::vector<double> tmp;
    DrainOnExit drain{c->stream};
    for (int t = 0; t < 3; ++t) {
      void* dv = nullptr;
      if ((rc = stage_beliefs(c, o, nv[t], kVdim[t], vhost[t], tmp, &dv, &used, arena, need))) return rc;
      st.nv[t] = nv[t]; st.bel[t] = (double*)dv;
    }
  }
  rome_upsolve_plan P;
  if ((rc = plan_build(c, &st, o, u, &P, true))) return rc;
  return plan_run(&P, o, nullptr, 0);
}

/* ---- native point containers <-> coordinates ---- */
int rome_points_to_coords(rome_ctx* c, int32_t dim, int32_t n, const double* pts, double* coords) {
  if (!c || n < 0 || (dim != 2 && dim != 3 && dim != 6) || (n > 0 && (!pts || !coords))) return ROME_ERR_INVALID_ARG;
  return convert_rows(c, dim, (size_t)n, pts, coords, true);
}
int rome_coords_to_points(rome_ctx* c, int32_t dim, int32_t n, const double* coords, double* pts) {
  if (!c || n < 0 || (dim != 2 && dim != 3 && dim != 6) || (n > 0 && (!pts || !coords))) return ROME_ERR_INVALID_ARG;
  return convert_rows(c, dim, (size_t)n, coords, pts, false);
}

/* ---- parametric linearisation ---- */
static bool lin_dims_host(int kind, int& dz, int& dr, int& da, int& db) {
  switch (kind) {
    case ROME_FACTOR_PRIORPOSE2: dz = 3; dr = 3; da = 3; db = 0; return true;
    case ROME_FACTOR_POSE2POSE2: dz = 3; dr = 3; da = 3; db = 3; return true;
    case ROME_FACTOR_POSE2POINT2BR: dz = 2; dr = 2; da = 3; db = 2; return true;
    case ROME_FACTOR_PRIORPOINT2: dz = 2; dr = 2; da = 2; db = 0; return true;
    case ROME_FACTOR_POSE3POSE3: dz = 6; dr = 6; da = 6; db = 6; return true;
    case ROME_FACTOR_PRIORPOSE3: dz = 6; dr = 6; da = 6; db = 0; return true;
    default: return false;
  }
}
int rome_linearize_dev(rome_ctx* c, int32_t kind, int32_t F, const double* mu, const double* W, const double* xa,
                       const double* xb, double* r, double* Ja, double* Jb) {
  int dz, dr, da, db;
  if (!c || F < 0 || !lin_dims_host(kind, dz, dr, da, db)) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  if (F > 0 && (!mu || !W || !xa || !r || !Ja || (db > 0 && (!xb || !Jb)))) return ROME_ERR_INVALID_ARG;
  ROME_HIP(c, rome::launch_linearize(kind, F, mu, W, xa, xb, r, Ja, Jb, c->stream));
  return ROME_OK;
}
int rome_linearize(rome_ctx* c, int32_t kind, int32_t F, const double* mu, const double* W, const double* xa,
                   const double* xb, double* r, double* Ja, double* Jb) {
  int dz, dr, da, db;
  if (!c || F < 0 || !lin_dims_host(kind, dz, dr, da, db)) return ROME_ERR_INVALID_ARG;
  if (F == 0) return ROME_OK;
  if (!mu || !W || !xa || !r || !Ja || (db > 0 && (!xb || !Jb))) return ROME_ERR_INVALID_ARG;
  ROME_HIP(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  void *d_mu, *d_W, *d_xa, *d_xb = nullptr, *d_r, *d_Ja, *d_Jb = nullptr;
  int rc;
  const size_t n = (size_t)F;
  if ((rc = ensure(c, 0, 8 * n * dz, &d_mu))) return rc;
  if ((rc = ensure(c, 1, 8 * n * dr * dr, &d_W))) return rc;
  if ((rc = ensure(c, 2, 8 * n * da, &d_xa))) return rc;
  if ((rc = ensure(c, 4, 8 * n * dr, &d_r))) return rc;
  if ((rc = ensure(c, 5, 8 * n * dr * da, &d_Ja))) return rc;
  ROME_HIP(c, hipMemcpyAsync(d_mu, mu, 8 * n * dz, hipMemcpyHostToDevice, s));
  ROME_HIP(c, hipMemcpyAsync(d_W, W, 8 * n * dr * dr, hipMemcpyHostToDevice, s));
  ROME_HIP(c, hipMemcpyAsync(d_xa, xa, 8 * n * da, hipMemcpyHostToDevice, s));
  if (db > 0) {
    if ((rc = ensure(c, 3, 8 * n * db, &d_xb))) return rc;
    if ((rc = ensure(c, 6, 8 * n * dr * db, &d_Jb))) return rc;
    ROME_HIP(c, hipMemcpyAsync(d_xb, xb, 8 * n * db, hipMemcpyHostToDevice, s));
  }
  ROME_HIP(c, rome::launch_linearize(kind, F, (const double*)d_mu, (const double*)d_W, (const double*)d_xa,
                                     (const double*)d_xb, (double*)d_r, (double*)d_Ja, (double*)d_Jb, s));
  ROME_HIP(c, hipMemcpyAsync(r, d_r, 8 * n * dr, hipMemcpyDeviceToHost, s));
  ROME_HIP(c, hipMemcpyAsync(Ja, d_Ja, 8 * n * dr * da, hipMemcpyDeviceToHost, s));
  if (db > 0) ROME_HIP(c, hipMemcpyAsync(Jb, d_Jb, 8 * n * dr * db, hipMemcpyDeviceToHost, s));
  ROME_HIP(c, hipStreamSynchronize(s));
  return ROME_OK;
}

/* ---- belief statistics / product ---- */
int rome_belief_stats_dev(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, double* mean, double* sd) {
  if (!c || V < 0 || N < 1 || (dim != 2 && dim != 3 && dim != 6) || (V > 0 && (!bel || !mean || !sd))) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  ROME_HIP(c, rome::launch_belief_stats(dim, V, N, bel, mean, sd, c->stream));
  return ROME_OK;
}
int rome_belief_stats(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, double* mean, double* sd) {
  if (!c || V < 0 || N < 1 || (dim != 2 && dim != 3 && dim != 6) || (V > 0 && (!bel || !mean || !sd))) return ROME_ERR_INVALID_ARG;
  if (V == 0) return ROME_OK;
  ROME_HIP(c, hipSetDevice(c->device));
  void *d_b, *d_m, *d_s; int rc;
  const size_t nb = 8ull * V * dim * N, nm = 8ull * V * dim;
  if ((rc = ensure(c, 0, nb, &d_b))) return rc;
  if ((rc = ensure(c, 1, nm, &d_m))) return rc;
  if ((rc = ensure(c, 2, nm, &d_s))) return rc;
  ROME_HIP(c, hipMemcpyAsync(d_b, bel, nb, hipMemcpyHostToDevice, c->stream));
  ROME_HIP(c, rome::launch_belief_stats(dim, V, N, (const double*)d_b, (double*)d_m, (double*)d_s, c->stream));
  ROME_HIP(c, hipMemcpyAsync(mean, d_m, nm, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipMemcpyAsync(sd, d_s, nm, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}
static int check_kde(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, const double* bw) {
  if (!c || V < 0 || N < 2 || N > ROME_MAX_PARTICLES_REGISTER || dim < 1 || dim > 6 || (V > 0 && (!bel || !bw))) return ROME_ERR_INVALID_ARG;
  return ROME_OK;
}
int rome_kde_bandwidth_dev(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, uint32_t circular_mask,
                           double tol_euclid, double tol_circular, double* bw) {
  int rc = check_kde(c, dim, V, N, bel, bw); if (rc) return rc;
  ROME_BIND(c);
  ROME_HIP(c, rome::launch_kde_bandwidth(dim, V, N, bel, circular_mask, tol_euclid > 0 ? tol_euclid : 1e-2,
                                         tol_circular > 0 ? tol_circular : 1e-6, bw, nullptr, c->stream));
  return ROME_OK;
}
int rome_kde_bandwidth(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, uint32_t circular_mask,
                       double tol_euclid, double tol_circular, double* bw) {
  int rc = check_kde(c, dim, V, N, bel, bw); if (rc) return rc;
  if (V == 0) return ROME_OK;
  ROME_HIP(c, hipSetDevice(c->device));
  void *d_b, *d_h;
  const size_t nb = 8ull * V * dim * N, nh = 8ull * V * dim;
  if ((rc = ensure(c, 0, nb, &d_b))) return rc;
  if ((rc = ensure(c, 1, nh, &d_h))) return rc;
  ROME_HIP(c, hipMemcpyAsync(d_b, bel, nb, hipMemcpyHostToDevice, c->stream));
  ROME_HIP(c, rome::launch_kde_bandwidth(dim, V, N, (const double*)d_b, circular_mask, tol_euclid > 0 ? tol_euclid : 1e-2,
                                         tol_circular > 0 ? tol_circular : 1e-6, (double*)d_h, nullptr, c->stream));
  ROME_HIP(c, hipMemcpyAsync(bw, d_h, nh, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}
int rome_kde_max_dev(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, const double* bw, int32_t grid_points,
                     double* out) {
  int rc = check_kde(c, dim, V, N, bel, bw); if (rc) return rc;
  ROME_BIND(c);
  const int G = grid_points > 0 ? grid_points : 200;
  if (G < 2 || G > 256 || (V > 0 && !out)) return ROME_ERR_INVALID_ARG;
  ROME_HIP(c, rome::launch_kde_max(dim, V, N, G, 0.1, bel, bw, out, c->stream));
  return ROME_OK;
}
int rome_kde_max(rome_ctx* c, int32_t dim, int32_t V, int32_t N, const double* bel, const double* bw, int32_t grid_points, double* out) {
  int rc = check_kde(c, dim, V, N, bel, bw); if (rc) return rc;
  const int G = grid_points > 0 ? grid_points : 200;
  if (G < 2 || G > 256 || (V > 0 && !out)) return ROME_ERR_INVALID_ARG;
  if (V == 0) return ROME_OK;
  ROME_HIP(c, hipSetDevice(c->device));
  void *d_b, *d_h, *d_o;
  const size_t nb = 8ull * V * dim * N, nh = 8ull * V * dim;
  if ((rc = ensure(c, 0, nb, &d_b))) return rc;
  if ((rc = ensure(c, 1, nh, &d_h))) return rc;
  if ((rc = ensure(c, 2, nh, &d_o))) return rc;
  ROME_HIP(c, hipMemcpyAsync(d_b, bel, nb, hipMemcpyHostToDevice, c->stream));
  ROME_HIP(c, hipMemcpyAsync(d_h, bw, nh, hipMemcpyHostToDevice, c->stream));
  ROME_HIP(c, rome::launch_kde_max(dim, V, N, G, 0.1, (const double*)d_b, (const double*)d_h, (double*)d_o, c->stream));
  ROME_HIP(c, hipMemcpyAsync(out, d_o, nh, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}
int rome_product_bw_dev(rome_ctx* c, const rome_opts* o, int32_t dim, int32_t V, const int32_t* prop_ptr, const int32_t* prop_rows,
                        const double* prop, const double* prop_bw, const double* bel_in, double* bel_out) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || V < 0 || (dim != 2 && dim != 3 && dim != 6)) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  if (V > 0 && (!prop_ptr || !bel_in || !bel_out)) return ROME_ERR_INVALID_ARG;
  const int N = o->n_particles;
  if (dim == 6 && N > 256) return ROME_ERR_UNSUPPORTED_N;   /* Pose3 product: points staged in LDS */
  const double c_n = std::pow(4.0 / ((dim + 2.0) * N), 1.0 / (dim + 4.0));
  ROME_HIP(c, rome::launch_product(dim, V, N, prop_ptr, prop_rows, prop, prop_bw, bel_in, bel_out, c_n, o->seed, o->stream_offset, c->stream));
  return ROME_OK;
}
int rome_product_dev(rome_ctx* c, const rome_opts* o, int32_t dim, int32_t V, const int32_t* prop_ptr, const int32_t* prop_rows,
                     const double* prop, const double* bel_in, double* bel_out) {
  return rome_product_bw_dev(c, o, dim, V, prop_ptr, prop_rows, prop, nullptr, bel_in, bel_out);
}

int rome_product_gibbs_dev(rome_ctx* c, const rome_opts* o, int32_t dim, int32_t V, const int32_t* prop_ptr, const int32_t* prop_rows,
                           const double* prop, const double* prop_bw, int32_t n_prop_rows, const double* bel_in, double* bel_out,
                           uint32_t circular_mask, int32_t gibbs_iters, int32_t max_proposals) {
  int rc = check_opts(o); if (rc) return rc;
  if (!c || V < 0 || n_prop_rows < 0 || (dim != 2 && dim != 3 && dim != 6) || max_proposals < 1) return ROME_ERR_INVALID_ARG;
  if (V > 0 && (!prop_ptr || !prop_rows || !bel_in || !bel_out)) return ROME_ERR_INVALID_ARG;
  if (n_prop_rows > 0 && (!prop || !prop_bw)) return ROME_ERR_INVALID_ARG;
  if (o->n_particles > ROME_MAX_PARTICLES_GIBBS) return ROME_ERR_UNSUPPORTED_N;   /* lane = output sample: 128- or 256-thread blocks */
  ROME_BIND(c);
  void* trees = nullptr;   /* one ball tree per proposal row, context-owned workspace (grown on demand, kept) */
  rc = ensure(c, 10, rome::gibbs_workspace_bytes(dim, n_prop_rows, V, o->n_particles), &trees); if (rc) return rc;
  ROME_HIP(c, rome::launch_product_gibbs(dim, V, o->n_particles, n_prop_rows, prop_ptr, prop_rows, prop, prop_bw, bel_in, bel_out, trees,
                                         circular_mask, gibbs_iters, max_proposals, o->seed, o->stream_offset, c->stream));
  return ROME_OK;
}

/* ---- device memory helpers ---- */
int rome_dev_alloc(rome_ctx* c, uint64_t bytes, void** out) {
  if (!c || !out) return ROME_ERR_INVALID_ARG;
  ROME_HIP(c, hipSetDevice(c->device));
  ROME_HIP(c, hipMalloc(out, bytes ? bytes : 8));
  return ROME_OK;
}
int rome_dev_free(rome_ctx* c, void* p) {
  if (!c) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  if (p) ROME_HIP(c, hipFree(p));
  return ROME_OK;
}
int rome_dev_upload(rome_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c || (bytes && (!dst || !src))) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  ROME_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}
int rome_dev_download(rome_ctx* c, void* dst, const void* src, uint64_t bytes) {
  if (!c || (bytes && (!dst || !src))) return ROME_ERR_INVALID_ARG;
  ROME_BIND(c);
  ROME_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  ROME_HIP(c, hipStreamSynchronize(c->stream));
  return ROME_OK;
}

}  // extern "C"
