// Net2DBillinear.sample_down = BilinearModule(3, 3, (384,384)) (models/image_models_billinear.py:8-24,41,131):
//   Conv1x1(3->3) -> ReLU -> BatchNorm2d(3) on the FULL-resolution image (train-mode statistics
//   need every pixel) -> nearest pick to oh x ow.
// Fused: one pass over the image for the statistics, one over the picked pixels for the output.
// The backward needs no full-resolution pass either: with v = relu(W x + b), m = (v > 0),
//   dW[o,c] = g*is*( sum_pix dy m x_c  -  S1/N * sum_pix m x_c  -  S2/N * sum_pix xhat m x_c )
// the last two sums do not depend on the incoming gradient and are produced by the forward
// statistics pass (xhat m x_c = is*(v x_c - mean m x_c)); only the picked pixels carry dy.
#include "ftx_common.h"

using namespace ftx;

constexpr int SD_NSUM = 27;   // v(3) v^2(3) m(3) m*x_c(9) v*x_c(9)
constexpr int SD_BLOCKS = 512;

__device__ inline int sd_nearest(int dst, int n_in, int n_out) {
  float scale = (float)n_in / (float)n_out;
  int s = (int)floorf((float)dst * scale);
  return s < n_in - 1 ? s : n_in - 1;
}

__device__ inline double block_sum(double v, double *sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
  return r;  // valid on thread 0
}

__global__ __launch_bounds__(256) void sd_stats_kernel(const float *__restrict__ img, int64_t hw, int b, const float *__restrict__ w9,
                                                       const float *__restrict__ b3, double *__restrict__ part) {
  __shared__ double sh[4];
  float W[9], B[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) W[i] = w9[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) B[i] = b3[i];
  double acc[SD_NSUM];
#pragma unroll
  for (int i = 0; i < SD_NSUM; ++i) acc[i] = 0;
  const int64_t total = (int64_t)b * hw;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t bi = p / hw, pix = p - bi * hw;
    const float *base = img + bi * 3 * hw + pix;
    float x[3] = {base[0], base[hw], base[2 * hw]};
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = W[o * 3] * x[0] + W[o * 3 + 1] * x[1] + W[o * 3 + 2] * x[2] + B[o];
      v = v > 0.f ? v : 0.f;
      float m = v > 0.f ? 1.f : 0.f;
      acc[o] += v;
      acc[3 + o] += (double)v * v;
      acc[6 + o] += m;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        acc[9 + o * 3 + c] += m * x[c];
        acc[18 + o * 3 + c] += (double)v * x[c];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < SD_NSUM; ++i) {
    double r = block_sum(acc[i], sh);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * SD_NSUM + i] = r;
  }
}

// tot[col] = sum over the nb partial rows, for NCOL <= 32 columns, by a block of 1024 threads: 32 lanes per column, each summing
// every 32nd row (independent loads in flight; one thread per column walking all rows took 117 us for 512 rows), then a fixed
// shuffle tree -- the order of the sum depends on nb only.  Ends with a block barrier.
template <int NCOL>
__device__ inline void sd_column_totals(const double *__restrict__ part, int nb, double *tot) {
  const int col = threadIdx.x >> 5, lane = threadIdx.x & 31;
  double s = 0;
  if (col < NCOL) {
#pragma unroll 8
    for (int k = lane; k < nb; k += 32) s += part[(int64_t)k * NCOL + col];
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
  if (col < NCOL && lane == 0) tot[col] = s;
  __syncthreads();
}

// saved[0..26] = totals, saved[27..29] = mean, saved[30..32] = invstd
__global__ __launch_bounds__(1024) void sd_finalize_fwd_kernel(const double *__restrict__ part, int nb, double n, float eps, float momentum, int training,
                                       float *__restrict__ running_mean, float *__restrict__ running_var, double *__restrict__ saved) {
  const int i = threadIdx.x;
  __shared__ double tot[SD_NSUM];
  sd_column_totals<SD_NSUM>(part, nb, tot);
  if (i < SD_NSUM) saved[i] = tot[i];
  if (i < 3) {
    double mean, var;
    if (training) {
      mean = tot[i] / n;
      var = tot[3 + i] / n - mean * mean;
      if (var < 0) var = 0;
      if (running_mean) running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * (float)mean;
      if (running_var) running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
    } else {
      mean = running_mean[i];
      var = running_var[i];
    }
    saved[27 + i] = mean;
    saved[30 + i] = 1.0 / sqrt(var + (double)eps);
  }
}

__global__ void sd_pick_kernel(const float *__restrict__ img, int b, int h, int w, int oh, int ow, const float *__restrict__ w9,
                               const float *__restrict__ b3, const float *__restrict__ gamma, const float *__restrict__ beta,
                               const double *__restrict__ saved, float *__restrict__ out) {
  const int64_t hw = (int64_t)h * w, ohw = (int64_t)oh * ow;
  const int64_t total = (int64_t)b * ohw;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t bi = p / ohw, q = p - bi * ohw;
    int oy = (int)(q / ow), ox = (int)(q - (int64_t)oy * ow);
    int64_t pix = (int64_t)sd_nearest(oy, h, oh) * w + sd_nearest(ox, w, ow);
    const float *base = img + bi * 3 * hw + pix;
    float x0 = base[0], x1 = base[hw], x2 = base[2 * hw];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = w9[o * 3] * x0 + w9[o * 3 + 1] * x1 + w9[o * 3 + 2] * x2 + b3[o];
      v = v > 0.f ? v : 0.f;
      out[(bi * 3 + o) * ohw + q] = (v - (float)saved[27 + o]) * (float)saved[30 + o] * gamma[o] + beta[o];
    }
  }
}

// backward sums over the picked pixels: [0..2] S1_o = sum dy, [3..5] S2_o = sum dy*xhat, [6..8] D_o = sum dy*m, [9..17] A_oc = sum dy*m*x_c
constexpr int SD_NBWD = 18;
__global__ __launch_bounds__(256) void sd_bwd_kernel(const float *__restrict__ img, const float *__restrict__ gy, int b, int h, int w, int oh,
                                                     int ow, const float *__restrict__ w9, const float *__restrict__ b3,
                                                     const double *__restrict__ saved, double *__restrict__ part) {
  __shared__ double sh[4];
  const int64_t hw = (int64_t)h * w, ohw = (int64_t)oh * ow;
  const int64_t total = (int64_t)b * ohw;
  double acc[SD_NBWD];
#pragma unroll
  for (int i = 0; i < SD_NBWD; ++i) acc[i] = 0;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t bi = p / ohw, q = p - bi * ohw;
    int oy = (int)(q / ow), ox = (int)(q - (int64_t)oy * ow);
    int64_t pix = (int64_t)sd_nearest(oy, h, oh) * w + sd_nearest(ox, w, ow);
    const float *base = img + bi * 3 * hw + pix;
    float x[3] = {base[0], base[hw], base[2 * hw]};
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      float v = w9[o * 3] * x[0] + w9[o * 3 + 1] * x[1] + w9[o * 3 + 2] * x[2] + b3[o];
      v = v > 0.f ? v : 0.f;
      float m = v > 0.f ? 1.f : 0.f;
      float dy = gy[(bi * 3 + o) * ohw + q];
      float xhat = (v - (float)saved[27 + o]) * (float)saved[30 + o];
      acc[o] += dy;
      acc[3 + o] += (double)dy * xhat;
      acc[6 + o] += dy * m;
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[9 + o * 3 + c] += (double)(dy * m) * x[c];
    }
  }
#pragma unroll
  for (int i = 0; i < SD_NBWD; ++i) {
    double r = block_sum(acc[i], sh);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.x * SD_NBWD + i] = r;
  }
}

__global__ __launch_bounds__(1024) void sd_finalize_bwd_kernel(const double *__restrict__ part, int nb, double n, const float *__restrict__ gamma,
                                       const double *__restrict__ saved, float *__restrict__ gw9, float *__restrict__ gb3,
                                       float *__restrict__ ggamma, float *__restrict__ gbeta) {
  __shared__ double t[SD_NBWD];
  const int i = threadIdx.x;
  sd_column_totals<SD_NBWD>(part, nb, t);
  if (i < 3) {
    const int o = i;
    const double mean = saved[27 + o], is = saved[30 + o], g = gamma[o];
    const double S1 = t[o], S2 = t[3 + o];
    gbeta[o] = (float)S1;
    ggamma[o] = (float)S2;
    // db = g*is*( D - S1/N * sum m - S2/N * sum xhat*m ),  sum xhat*m = is*(sum v - mean * sum m)
    const double sum_m = saved[6 + o], sum_v = saved[o];
    gb3[o] = (float)(g * is * (t[6 + o] - S1 / n * sum_m - S2 / n * is * (sum_v - mean * sum_m)));
    for (int c = 0; c < 3; ++c) {
      const double mx = saved[9 + o * 3 + c], vx = saved[18 + o * 3 + c];
      gw9[o * 3 + c] = (float)(g * is * (t[9 + o * 3 + c] - S1 / n * mx - S2 / n * is * (vx - mean * mx)));
    }
  }
}

extern "C" size_t ftx_sample_down_workspace_bytes(void) { return sizeof(double) * SD_BLOCKS * SD_NSUM + 256; }

extern "C" int ftx_sample_down_fwd(const float *img, int32_t b, int32_t h, int32_t w, int32_t oh, int32_t ow, const float *conv_w, const float *conv_b,
                                   const float *gamma, const float *beta, float *running_mean, float *running_var, float momentum, float eps,
                                   int32_t training, float *out, double *saved, void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(b >= 1 && h >= 1 && w >= 1 && oh >= 1 && ow >= 1, "ftx_sample_down_fwd: bad size");
  FTX_REQUIRE(img && conv_w && conv_b && gamma && beta && out && saved && workspace, "ftx_sample_down_fwd: null pointer");
  FTX_REQUIRE(training || (running_mean && running_var), "ftx_sample_down_fwd: eval mode needs running statistics");
  if (workspace_bytes < ftx_sample_down_workspace_bytes()) {
    set_error("ftx_sample_down_fwd: workspace too small");
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)workspace;
  const int64_t hw = (int64_t)h * w;
  sd_stats_kernel<<<SD_BLOCKS, 256, 0, st>>>(img, hw, b, conv_w, conv_b, part);
  sd_finalize_fwd_kernel<<<1, 1024, 0, st>>>(part, SD_BLOCKS, (double)b * hw, eps, momentum, training, running_mean, running_var, saved);
  sd_pick_kernel<<<grid_for((int64_t)b * oh * ow, 256), 256, 0, st>>>(img, b, h, w, oh, ow, conv_w, conv_b, gamma, beta, saved, out);
  return check_launch("ftx_sample_down_fwd");
}

extern "C" int ftx_sample_down_bwd(const float *img, const float *grad_out, int32_t b, int32_t h, int32_t w, int32_t oh, int32_t ow,
                                   const float *conv_w, const float *conv_b, const float *gamma, const double *saved, float *grad_conv_w,
                                   float *grad_conv_b, float *grad_gamma, float *grad_beta, void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(b >= 1 && h >= 1 && w >= 1 && oh >= 1 && ow >= 1, "ftx_sample_down_bwd: bad size");
  FTX_REQUIRE(img && grad_out && conv_w && conv_b && gamma && saved && grad_conv_w && grad_conv_b && grad_gamma && grad_beta && workspace,
              "ftx_sample_down_bwd: null pointer");
  if (workspace_bytes < ftx_sample_down_workspace_bytes()) {
    set_error("ftx_sample_down_bwd: workspace too small");
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)workspace;
  sd_bwd_kernel<<<SD_BLOCKS, 256, 0, st>>>(img, grad_out, b, h, w, oh, ow, conv_w, conv_b, saved, part);
  sd_finalize_bwd_kernel<<<1, 1024, 0, st>>>(part, SD_BLOCKS, (double)b * h * w, gamma, saved, grad_conv_w, grad_conv_b, grad_gamma, grad_beta);
  return check_launch("ftx_sample_down_bwd");
}
