// BatchNorm1d over the rows of a voxel/point feature matrix, fused with the optional
// residual add and ReLU that follow it in every SPVCNN block
// (models/spvcnn.py:30-31,71-79,100-102,164-180).  HBM-bound: x is read twice in the
// forward (statistics, apply) and the output written once; statistics accumulate in
// float64 so mean/var do not depend on how rows are split over workgroups.
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include "ftx_common.h"
#include "ftx_lastblock.h"

using namespace ftx;

// ---- per-(device, stream) ticket buffer (ftx_lastblock.h) ------------------------------------------------------------------------
// The statistics kernels hand their column totals to "the last block to finish"; the tickets and the group rows live in a small
// buffer that belongs to ONE stream of ONE device (launches of a stream are serialised, the last user of a counter puts it back to
// zero).  The CALLER can own it: ftx_stream_scratch_bytes() / ftx_stream_scratch_attach(stream, ptr, bytes) hand the library a buffer
// of the caller's, ftx_stream_scratch_reset(stream) clears the tickets (after a kernel died mid-flight), ftx_stream_scratch_release
// gives it back.  Only a stream nobody attached a buffer to gets a library allocation, at first use (freed by release).
namespace ftx {
namespace {
struct ScratchEntry {
  StreamScratch sc;
  void *owned;      // non-null: allocated here (hipMalloc), freed by release
};
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, ScratchEntry> g_scratch;

constexpr size_t scratch_counter_bytes() { return 512 * ((sizeof(uint32_t) * (1 + LB_MAX_GROUPS) + 511) / 512); }
constexpr size_t scratch_group_bytes() { return sizeof(double) * (size_t)LB_MAX_GROUPS * 2 * LB_MAX_COLS; }

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) (void)hipGetLastError();
  return dev;
}
}  // namespace

StreamScratch stream_scratch(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  const auto key = std::make_pair(current_device(), st);     // the legacy stream has the same handle on every device
  auto it = g_scratch.find(key);
  if (it != g_scratch.end()) return it->second.sc;
  StreamScratch sc{nullptr, nullptr};
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
    set_error("stream_scratch: first use of a stream inside a capture (attach a buffer, or run the op once on that stream, before capturing)");
    return sc;
  }
  (void)hipGetLastError();
  char *base = nullptr;
  if (hipMalloc((void **)&base, scratch_counter_bytes() + scratch_group_bytes()) != hipSuccess ||
      hipMemsetAsync(base, 0, scratch_counter_bytes(), st) != hipSuccess) {
    set_error("stream_scratch: cannot allocate the per-stream ticket buffer (%s)", hipGetErrorString(hipGetLastError()));
    if (base) (void)hipFree(base);
    return sc;
  }
  sc.counters = (uint32_t *)base;
  sc.gpart = (double *)(base + scratch_counter_bytes());
  g_scratch[key] = ScratchEntry{sc, base};
  return sc;
}
}  // namespace ftx

extern "C" size_t ftx_stream_scratch_bytes(void) { return scratch_counter_bytes() + scratch_group_bytes(); }

extern "C" int ftx_stream_scratch_attach(void *stream, void *buffer, size_t bytes) {
  FTX_REQUIRE(buffer && ((uintptr_t)buffer & 255) == 0, "ftx_stream_scratch_attach: the buffer must be non-null and 256-byte aligned");
  FTX_REQUIRE(bytes >= ftx_stream_scratch_bytes(), "ftx_stream_scratch_attach: %zu bytes < ftx_stream_scratch_bytes() = %zu", bytes, ftx_stream_scratch_bytes());
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  const auto key = std::make_pair(current_device(), st);
  if (hipMemsetAsync(buffer, 0, scratch_counter_bytes(), st) != hipSuccess) return check_launch("ftx_stream_scratch_attach memset");
  auto it = g_scratch.find(key);
  if (it != g_scratch.end() && it->second.owned) {
    // kernels already queued on the stream may still use the library's buffer: free it when they are done
    if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();
    (void)hipFree(it->second.owned);
  }
  StreamScratch sc{(uint32_t *)buffer, (double *)((char *)buffer + scratch_counter_bytes())};
  g_scratch[key] = ScratchEntry{sc, nullptr};
  return FTX_OK;
}

extern "C" int ftx_stream_scratch_reset(void *stream) {
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto it = g_scratch.find(std::make_pair(current_device(), st));
  if (it == g_scratch.end()) return FTX_OK;      // nothing to clear
  if (hipMemsetAsync(it->second.sc.counters, 0, scratch_counter_bytes(), st) != hipSuccess) return check_launch("ftx_stream_scratch_reset");
  return FTX_OK;
}

extern "C" int ftx_stream_scratch_release(void *stream) {
  hipStream_t st = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto it = g_scratch.find(std::make_pair(current_device(), st));
  if (it == g_scratch.end()) return FTX_OK;
  if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();      // the buffer goes back to its owner: nothing may still use it
  if (it->second.owned) (void)hipFree(it->second.owned);
  g_scratch.erase(it);
  return FTX_OK;
}

// Grid of the statistics passes.  They stream 1-3 row matrices once and are bound by loads in flight, not by bytes: at 128 rows per
// block the 81k-row level ran 635 blocks (2.5 per CU) and reached 1.7 TB/s.  48 rows per block, at most 2048 blocks: constants, because
// the grid fixes the summation tree and with it the bits of the statistics (sweeps: DESIGN.md section 5, rounds 2 and 3).
static int bn_blocks(int64_t n) {
  constexpr int rows = 48, maxb = 2048;
  int64_t b = ceil_div(n, rows);
  if (b > maxb) b = maxb;
  if (b > LB_GROUP * LB_MAX_GROUPS) b = LB_GROUP * LB_MAX_GROUPS;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t ftx_bn_workspace_bytes(int64_t n, int32_t c) {
  if (c <= 0) return 256;
  return sizeof(double) * (size_t)bn_blocks(n) * 2 * c + sizeof(double) * 2 * c;
}

// Per-block partial column sums of two quantities (q0, q1) produced by `Op` for each element.
//   forward:  q0 = x,        q1 = x*x
//   backward: q0 = dy,       q1 = dy * xhat      (dy masked by y>0 when relu)
// y = (x - mean) * invstd * gamma + beta with the rounding of every step pinned (no contraction left to the compiler): the forward apply
// pass and the ReLU mask the backward RECOMPUTES from x (instead of reading y back: a third of the backward's traffic for the
// BatchNorm + ReLU layers that have no residual) must agree to the last bit.
__device__ inline float bn_affine(float x, float mean, float invstd, float gamma, float beta) {
  return __fmaf_rn(__fmul_rn(__fsub_rn(x, mean), invstd), gamma, beta);
}

struct FwdOp {
  __device__ static void eval(float x, float, float, float, float, float, float, int, double &q0, double &q1) {
    q0 = (double)x;
    q1 = (double)x * (double)x;
  }
};
// relu: 0 none, 1 mask from the stored forward output y, 2 mask recomputed from x (no residual went into y)
struct BwdOp {
  __device__ static void eval(float x, float gy, float y, float mean, float invstd, float gamma, float beta, int relu, double &q0, double &q1) {
    if (relu == 2) y = bn_affine(x, mean, invstd, gamma, beta);
    float dy = (relu && !(y > 0.f)) ? 0.f : gy;
    q0 = (double)dy;
    q1 = (double)dy * (double)((x - mean) * invstd);
  }
};

template <class Op>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                         const float *__restrict__ y, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, int relu, int64_t n, int c,
                                                         double *part, StreamScratch sc) {
  // part: gridDim.x rows [2][c], then the totals row [2][c] written by the last block to finish (ftx_lastblock.h)
  extern __shared__ double sh[];  // [2][RL][c]
  const int c4 = c >> 2;
  const int RL = 256 / c4 > 0 ? 256 / c4 : 1;
  const int tid = threadIdx.x;
  const int cg = tid % c4, rl = tid / c4;
  const int64_t rows_per_block = ceil_div(n, (int64_t)gridDim.x);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (rl < RL) {
    float4 mu = make_float4(0, 0, 0, 0), is = mu, gm = mu, bt = mu;
    if (mean) {
      mu = *(const float4 *)&mean[cg * 4];
      is = *(const float4 *)&invstd[cg * 4];
    }
    if (relu == 2) {
      gm = *(const float4 *)&gamma[cg * 4];
      bt = *(const float4 *)&beta[cg * 4];
    }
    auto accumulate = [&](const float4 &xv, const float4 &gv, const float4 &yv) {
      double a, b;
      Op::eval(xv.x, gv.x, yv.x, mu.x, is.x, gm.x, bt.x, relu, a, b); s0[0] += a; s1[0] += b;
      Op::eval(xv.y, gv.y, yv.y, mu.y, is.y, gm.y, bt.y, relu, a, b); s0[1] += a; s1[1] += b;
      Op::eval(xv.z, gv.z, yv.z, mu.z, is.z, gm.z, bt.z, relu, a, b); s0[2] += a; s1[2] += b;
      Op::eval(xv.w, gv.w, yv.w, mu.w, is.w, gm.w, bt.w, relu, a, b); s0[3] += a; s1[3] += b;
    };
    const float4 zero4 = make_float4(0, 0, 0, 0);
    int64_t r = r0 + rl;
    for (; r + RL < r1; r += 2 * RL) {   // two rows per trip: six independent 16-byte loads in flight per thread, summed in row order
      const int64_t o0 = r * c + cg * 4, o1 = (r + RL) * c + cg * 4;
      const float4 xa = *(const float4 *)&x[o0], xb = *(const float4 *)&x[o1];
      const float4 ga = gy ? *(const float4 *)&gy[o0] : zero4, gb = gy ? *(const float4 *)&gy[o1] : zero4;
      const float4 ya = y ? *(const float4 *)&y[o0] : zero4, yb = y ? *(const float4 *)&y[o1] : zero4;
      accumulate(xa, ga, ya);
      accumulate(xb, gb, yb);
    }
    for (; r < r1; r += RL) {
      const int64_t o0 = r * c + cg * 4;
      accumulate(*(const float4 *)&x[o0], gy ? *(const float4 *)&gy[o0] : zero4, y ? *(const float4 *)&y[o0] : zero4);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      sh[(0 * RL + rl) * c + cg * 4 + v] = s0[v];
      sh[(1 * RL + rl) * c + cg * 4 + v] = s1[v];
    }
  }
  __syncthreads();
  for (int j = tid; j < 2 * c; j += 256) {
    int which = j / c, col = j - which * c;
    double s = 0;
    for (int q = 0; q < RL; ++q) s += sh[(which * RL + q) * c + col];
    lb_store(&part[((int64_t)blockIdx.x * 2 + which) * c + col], s);
  }
  __syncthreads();   // sh is free again: 2 * RL * c = 2048 doubles >= 256 + 2c for c <= 512
  last_block_totals(part, (int)gridDim.x, c, sc, sh, StoreTotals{part + (int64_t)gridDim.x * 2 * c, c});
}

// mean / invstd of a column from its two totals, in float64 (every apply block computes the same values; block 0 also stores them
// and updates the running statistics -- the work of the former finalize launch).
__device__ inline void bn_column_stats(double s, double ss, int64_t n, float eps, float &mean, float &invstd, double &var_out) {
  const double m = s / (double)n;
  double var = ss / (double)n - m * m;
  if (var < 0) var = 0;
  mean = (float)m;
  invstd = (float)(1.0 / sqrt(var + (double)eps));
  var_out = var;
}

// Apply passes: thread (column group cg, row lane rl) keeps its four channels' constants in registers and walks rows rl, rl + RL*grid, ...
// two at a time (independent 16-byte loads in flight); RL = 256 / (c/4) rows per block and trip.
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const double *__restrict__ totals, float eps,
                                                           float momentum, float *__restrict__ running_mean, float *__restrict__ running_var,
                                                           float *__restrict__ save_mean, float *__restrict__ save_invstd, int64_t n, int c,
                                                           int relu, float *__restrict__ y) {
  const int c4 = c >> 2;
  const int RL = 256 / c4 > 0 ? 256 / c4 : 1;
  const int cg = threadIdx.x % c4, rl = threadIdx.x / c4;
  if (rl >= RL) return;
  const int col = cg * 4;
  float mu_[4], is_[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    double var;
    bn_column_stats(totals[col + v], totals[c + col + v], n, eps, mu_[v], is_[v], var);
    if (blockIdx.x == 0 && rl == 0) {
      save_mean[col + v] = mu_[v];
      save_invstd[col + v] = is_[v];
      if (running_mean) running_mean[col + v] = (1.f - momentum) * running_mean[col + v] + momentum * mu_[v];
      if (running_var) {
        const double unbiased = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_var[col + v] = (1.f - momentum) * running_var[col + v] + momentum * (float)unbiased;
      }
    }
  }
  const float4 mu = make_float4(mu_[0], mu_[1], mu_[2], mu_[3]), is = make_float4(is_[0], is_[1], is_[2], is_[3]);
  const float4 g = *(const float4 *)&gamma[col], b = *(const float4 *)&beta[col];
  auto one = [&](int64_t r) {
    const int64_t o = r * c + col;
    const float4 xv = *(const float4 *)&x[o];
    float4 ov;
    ov.x = bn_affine(xv.x, mu.x, is.x, g.x, b.x);
    ov.y = bn_affine(xv.y, mu.y, is.y, g.y, b.y);
    ov.z = bn_affine(xv.z, mu.z, is.z, g.z, b.z);
    ov.w = bn_affine(xv.w, mu.w, is.w, g.w, b.w);
    if (res) {
      const float4 rv = *(const float4 *)&res[o];
      ov.x += rv.x; ov.y += rv.y; ov.z += rv.z; ov.w += rv.w;
    }
    if (relu) {
      ov.x = ov.x > 0.f ? ov.x : 0.f; ov.y = ov.y > 0.f ? ov.y : 0.f;
      ov.z = ov.z > 0.f ? ov.z : 0.f; ov.w = ov.w > 0.f ? ov.w : 0.f;
    }
    *(float4 *)&y[o] = ov;
  };
  const int64_t stride = (int64_t)gridDim.x * RL;
  int64_t r = (int64_t)blockIdx.x * RL + rl;
  for (; r + stride < n; r += 2 * stride) {
    one(r);
    one(r + stride);
  }
  if (r < n) one(r);
}

static unsigned bn_apply_grid(int64_t n, int c) {
  const int c4 = c / 4;
  const int rl = 256 / c4 > 0 ? 256 / c4 : 1;
  int64_t g = ceil_div(n, 2 * (int64_t)rl);   // two rows per thread
  if (g > 4096) g = 4096;
  return (unsigned)(g < 1 ? 1 : g);
}

static int bn_check(const char *who, int64_t n, int c) {
  FTX_REQUIRE(n >= 0, "%s: n < 0", who);
  FTX_REQUIRE(c >= 4 && c % 4 == 0 && c <= 1024, "%s: c must be a multiple of 4 in [4,1024] (got %d)", who, c);
  return FTX_OK;
}

static size_t bn_partial_lds(int c) {
  int c4 = c / 4;
  int RL = 256 / c4 > 0 ? 256 / c4 : 1;
  return sizeof(double) * 2 * RL * c;
}

extern "C" int ftx_bn_train_fwd(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean,
                                float *running_var, float momentum, float eps, int64_t n, int32_t c, int32_t relu, float *y,
                                float *save_mean, float *save_invstd, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = bn_check("ftx_bn_train_fwd", n, c);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(n >= 1, "ftx_bn_train_fwd: needs at least one row");
  FTX_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && workspace, "ftx_bn_train_fwd: null pointer");
  FTX_REQUIRE(c <= 512, "ftx_bn_train_fwd: c > 512 unsupported");
  if (workspace_bytes < ftx_bn_workspace_bytes(n, c)) {
    set_error("ftx_bn_train_fwd: workspace %zu < required %zu", workspace_bytes, ftx_bn_workspace_bytes(n, c));
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const StreamScratch sc = stream_scratch(st);
  if (!sc.counters) return FTX_ELAUNCH;
  const int nb = bn_blocks(n);
  double *part = (double *)workspace;
  bn_partial_kernel<FwdOp><<<nb, 256, bn_partial_lds(c), st>>>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, n, c, part, sc);
  bn_apply_fwd_kernel<<<bn_apply_grid(n, c), 256, 0, st>>>(x, residual, gamma, beta, part + (size_t)nb * 2 * c, eps, momentum, running_mean,
                                                           running_var, save_mean, save_invstd, n, c, relu, y);
  return check_launch("ftx_bn_train_fwd");
}

// BatchNorm forward whose statistics were produced by the pass that wrote x (ftx_spconv_reduce_stats): `totals` is the row
// [2][c] of float64 column sums / sums of squares that pass leaves behind its partial rows.  One launch: x is read once.
extern "C" int ftx_bn_train_fwd_totals(const float *x, const float *residual, const float *gamma, const float *beta, float *running_mean,
                                       float *running_var, float momentum, float eps, int64_t n, int32_t c, int32_t relu, float *y,
                                       float *save_mean, float *save_invstd, const double *totals, void *stream) {
  int rc = bn_check("ftx_bn_train_fwd_totals", n, c);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(n >= 1, "ftx_bn_train_fwd_totals: needs at least one row");
  FTX_REQUIRE(x && gamma && beta && y && save_mean && save_invstd && totals, "ftx_bn_train_fwd_totals: null pointer");
  bn_apply_fwd_kernel<<<bn_apply_grid(n, c), 256, 0, (hipStream_t)stream>>>(x, residual, gamma, beta, totals, eps, momentum, running_mean,
                                                                            running_var, save_mean, save_invstd, n, c, relu, y);
  return check_launch("ftx_bn_train_fwd_totals");
}

__global__ void bn_apply_eval_kernel(const float *__restrict__ x, const float *__restrict__ res, const float *__restrict__ gamma,
                                     const float *__restrict__ beta, const float *__restrict__ rm, const float *__restrict__ rv, float eps,
                                     int64_t n, int c, int relu, float *__restrict__ y) {
  const int c4 = c >> 2;
  const int64_t total = n * c4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int col = (int)(e % c4) * 4;
    float4 xv = *(const float4 *)&x[e * 4];
    float o[4] = {xv.x, xv.y, xv.z, xv.w};
    float rr[4] = {0, 0, 0, 0};
    if (res) {
      float4 t = *(const float4 *)&res[e * 4];
      rr[0] = t.x; rr[1] = t.y; rr[2] = t.z; rr[3] = t.w;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float is = (float)(1.0 / sqrt((double)rv[col + v] + (double)eps));
      float t = (o[v] - rm[col + v]) * is * gamma[col + v] + beta[col + v] + rr[v];
      o[v] = (relu && !(t > 0.f)) ? 0.f : t;
    }
    *(float4 *)&y[e * 4] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

extern "C" int ftx_bn_eval_fwd(const float *x, const float *residual, const float *gamma, const float *beta, const float *running_mean,
                               const float *running_var, float eps, int64_t n, int32_t c, int32_t relu, float *y, void *stream) {
  int rc = bn_check("ftx_bn_eval_fwd", n, c);
  if (rc != FTX_OK) return rc;
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(x && gamma && beta && running_mean && running_var && y, "ftx_bn_eval_fwd: null pointer");
  bn_apply_eval_kernel<<<grid_for(n * (c / 4), 256), 256, 0, (hipStream_t)stream>>>(x, residual, gamma, beta, running_mean, running_var, eps,
                                                                                     n, c, relu, y);
  return check_launch("ftx_bn_eval_fwd");
}

__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ x, const float *__restrict__ y,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const double *__restrict__ sums, int64_t n, int c,
                                                           int relu, float *__restrict__ gx, float *__restrict__ gres,
                                                           float *__restrict__ grad_gamma, float *__restrict__ grad_beta) {
  const int c4 = c >> 2;
  const int RL = 256 / c4 > 0 ? 256 / c4 : 1;
  const int cg = threadIdx.x % c4, rl = threadIdx.x / c4;
  if (rl >= RL) return;
  const int col = cg * 4;
  const float inv_n = 1.f / (float)n;
  float mu[4], is[4], gm[4], bt[4], sdy[4], sdyx[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    mu[v] = mean[col + v];
    is[v] = invstd[col + v];
    gm[v] = gamma[col + v];
    bt[v] = relu == 2 ? beta[col + v] : 0.f;
    sdy[v] = (float)sums[col + v] * inv_n;
    sdyx[v] = (float)sums[c + col + v] * inv_n;
    if (blockIdx.x == 0 && rl == 0) {   // the parameter gradients are the two totals themselves
      if (grad_beta) grad_beta[col + v] = (float)sums[col + v];
      if (grad_gamma) grad_gamma[col + v] = (float)sums[c + col + v];
    }
  }
  auto one = [&](int64_t r) {
    const int64_t o = r * c + col;
    const float4 g4 = *(const float4 *)&gy[o];
    const float4 x4 = *(const float4 *)&x[o];
    float dy[4] = {g4.x, g4.y, g4.z, g4.w};
    const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
    if (relu == 2) {          // the forward output recomputed from x (bit for bit: bn_affine), not read back
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (!(bn_affine(xv[v], mu[v], is[v], gm[v], bt[v]) > 0.f)) dy[v] = 0.f;
    } else if (relu) {
      const float4 y4 = *(const float4 *)&y[o];
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (!(yv[v] > 0.f)) dy[v] = 0.f;
    }
    float ov[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float xhat = (xv[v] - mu[v]) * is[v];
      ov[v] = gm[v] * is[v] * (dy[v] - sdy[v] - xhat * sdyx[v]);
    }
    *(float4 *)&gx[o] = make_float4(ov[0], ov[1], ov[2], ov[3]);
    if (gres) *(float4 *)&gres[o] = make_float4(dy[0], dy[1], dy[2], dy[3]);
  };
  const int64_t stride = (int64_t)gridDim.x * RL;
  int64_t r = (int64_t)blockIdx.x * RL + rl;
  for (; r + stride < n; r += 2 * stride) {
    one(r);
    one(r + stride);
  }
  if (r < n) one(r);
}

extern "C" int ftx_bn_train_bwd(const float *grad_y, const float *x, const float *y, const float *gamma, const float *beta, const float *save_mean,
                                const float *save_invstd, int64_t n, int32_t c, int32_t relu, float *grad_x, float *grad_residual,
                                float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = bn_check("ftx_bn_train_bwd", n, c);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(n >= 1, "ftx_bn_train_bwd: needs at least one row");
  FTX_REQUIRE(grad_y && x && gamma && save_mean && save_invstd && grad_x && workspace, "ftx_bn_train_bwd: null pointer");
  // ReLU mask: recomputed from x when the forward had no residual and beta is given (y is then not read and may be NULL), else from y
  const bool remask = relu && beta != nullptr && grad_residual == nullptr;
  FTX_REQUIRE(!relu || remask || y, "ftx_bn_train_bwd: relu needs the forward output y (or beta, when no residual went into it)");
  const int rmode = !relu ? 0 : (remask ? 2 : 1);
  FTX_REQUIRE(c <= 512, "ftx_bn_train_bwd: c > 512 unsupported");
  if (workspace_bytes < ftx_bn_workspace_bytes(n, c)) {
    set_error("ftx_bn_train_bwd: workspace %zu < required %zu", workspace_bytes, ftx_bn_workspace_bytes(n, c));
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const StreamScratch sc = stream_scratch(st);
  if (!sc.counters) return FTX_ELAUNCH;
  const int nb = bn_blocks(n);
  double *part = (double *)workspace;
  double *sums = part + (size_t)nb * 2 * c;   // written by the last block of the statistics pass
  bn_partial_kernel<BwdOp><<<nb, 256, bn_partial_lds(c), st>>>(x, grad_y, rmode == 1 ? y : nullptr, save_mean, save_invstd, gamma, beta, rmode, n, c, part, sc);
  bn_apply_bwd_kernel<<<bn_apply_grid(n, c), 256, 0, st>>>(grad_y, x, y, gamma, beta, save_mean, save_invstd, sums, n, c, rmode, grad_x, grad_residual,
                                                           grad_gamma, grad_beta);
  return check_launch("ftx_bn_train_bwd");
}

// ---------------------------------------------------------------------------------------
// Column sums of a row-major (rows, cols) float32 matrix: the bias gradient of every Linear of the ViT trunk and of the point
// branch (the reference reaches it through autograd's sum_to reduction: models/transformers.py:16-45, timm Mlp / Attention).
// Pass 1: block = up to 256 columns (<= 64 lanes x float4 of a row) x (256 / lanes) row lanes over one chunk of rows -> one float64 partial row;
// pass 2: 16 lanes per column sum the chunk rows (every 16th each, combined in lane order).  Float64 throughout: the rounding to float32 happens once.
// A (1, M) x (M, N) library GEMM took 14 us for the 2312 x 768..3072 gradients, 48 times per step; these two take 3-6 us together.
// ---------------------------------------------------------------------------------------
// geometry of pass 1: c4w float4 column groups per row (<= 64: 1 KB of a row per wave), RL = 256 / c4w row lanes, `chunks` row ranges
struct ColsumGeom {
  int c4w, rl, slab, chunks;
};
static ColsumGeom colsum_geom(int64_t rows, int cols) {
  ColsumGeom g;
  const int c4 = (cols + 3) / 4;
  g.c4w = c4 < 64 ? c4 : 64;
  g.rl = 256 / g.c4w;
  g.slab = g.c4w * 4;
  int64_t ch = ceil_div(rows, (int64_t)g.rl * 8);   // ~8 rows per thread
  if (ch > 256) ch = 256;
  g.chunks = (int)(ch < 1 ? 1 : ch);
  return g;
}
extern "C" size_t ftx_colsum_workspace_bytes(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 256;
  return sizeof(double) * (size_t)colsum_geom(rows, cols).chunks * (size_t)cols + 256;
}

__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ x, int64_t rows, int cols, int c4w, int RL,
                                                             double *__restrict__ part) {
  __shared__ double sh[1024];   // [RL][slab], RL * slab <= 256 * 4
  const int slab = c4w * 4;
  const int cq = threadIdx.x % c4w, rl = threadIdx.x / c4w;
  const int col = blockIdx.x * slab + cq * 4;
  const int64_t per = ceil_div(rows, (int64_t)gridDim.y);
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  if (rl < RL) {
    double s[4] = {0, 0, 0, 0};
    if (col < cols) {
      int64_t r = r0 + rl;
      for (; r + RL < r1; r += 2 * RL) {   // two rows in flight per thread
        const float4 a = *(const float4 *)&x[r * cols + col], b = *(const float4 *)&x[(r + RL) * cols + col];
        s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
        s[0] += (double)b.x; s[1] += (double)b.y; s[2] += (double)b.z; s[3] += (double)b.w;
      }
      for (; r < r1; r += RL) {
        const float4 a = *(const float4 *)&x[r * cols + col];
        s[0] += (double)a.x; s[1] += (double)a.y; s[2] += (double)a.z; s[3] += (double)a.w;
      }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) sh[rl * slab + cq * 4 + v] = s[v];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < slab; j += 256) {
    const int c = blockIdx.x * slab + j;
    if (c < cols) {
      double t = 0;
      for (int q = 0; q < RL; ++q) t += sh[q * slab + j];
      part[(int64_t)blockIdx.y * cols + c] = t;
    }
  }
}

// pass 2: block = 16 columns x 16 chunk lanes; a lane sums every 16th partial row (independent loads), the 16 lanes are combined in order
__global__ __launch_bounds__(256) void colsum_final_kernel(const double *__restrict__ part, int chunks, int cols, float *__restrict__ out) {
  __shared__ double sh[16][16];
  const int cw = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cw;
  double s = 0;
  if (c < cols) {
#pragma unroll 4
    for (int k = cl; k < chunks; k += 16) s += part[(int64_t)k * cols + c];
  }
  sh[cl][cw] = s;
  __syncthreads();
  if (cl == 0 && c < cols) {
    double t = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][cw];
    out[c] = (float)t;
  }
}

extern "C" int ftx_colsum(const float *x, int64_t rows, int32_t cols, float *out, void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(rows >= 0 && cols >= 4 && cols % 4 == 0, "ftx_colsum: cols must be a positive multiple of 4 (got %d)", cols);
  FTX_REQUIRE(out && workspace && (x || rows == 0), "ftx_colsum: null pointer");
  if (workspace_bytes < ftx_colsum_workspace_bytes(rows, cols)) {
    set_error("ftx_colsum: workspace %zu < required %zu", workspace_bytes, ftx_colsum_workspace_bytes(rows, cols));
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const ColsumGeom g = colsum_geom(rows, cols);
  double *part = (double *)workspace;
  dim3 grid((unsigned)ceil_div(cols, g.slab), (unsigned)g.chunks);
  colsum_partial_kernel<<<grid, 256, 0, st>>>(x, rows, cols, g.c4w, g.rl, part);
  colsum_final_kernel<<<(unsigned)ceil_div(cols, 16), 256, 0, st>>>(part, g.chunks, cols, out);
  return check_launch("ftx_colsum");
}
