// Point <-> voxel feature movement (spvoxelize / spdevoxelize fwd+bwd), the fused
// nearest-upsample + per-point lift gather, and the NCHW nearest resample.
// HBM-bound row gathers/scatters: one thread moves 16 bytes of one row, lanes of a
// wave cover consecutive 16-byte pieces of the same row (coalesced 128..1024 B).
#include "ftx_common.h"

using namespace ftx;

// ---------------------------------------------------------------- voxelize
// out[idx[i], :] += feats[i, :] / counts[idx[i]]        (scatter-mean, float atomics)
template <int VEC>
__global__ void voxelize_fwd_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx, const int32_t *__restrict__ counts,
                                    int64_t n, int c, int64_t m, float *__restrict__ out) {
  const int cv = c / VEC;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * VEC;
    int32_t pos = idx[i];
    if (pos < 0 || pos >= m) continue;
    int32_t cnt = counts[pos];
    if (cnt == 0) continue;
    const float fc = (float)cnt;
#pragma unroll
    for (int v = 0; v < VEC; ++v) atomicAdd(&out[(int64_t)pos * c + j + v], feats[i * c + j + v] / fc);
  }
}

extern "C" int ftx_voxelize_fwd(const float *feats, const int32_t *idx, const int32_t *counts, int64_t n, int32_t c, int64_t m,
                                float *out, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 1, "ftx_voxelize_fwd: bad size");
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(out && counts && ((feats && idx) || n == 0), "ftx_voxelize_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float) * m * c, st) != hipSuccess) return check_launch("ftx_voxelize_fwd memset");
  if (n == 0) return FTX_OK;
  if (c % 4 == 0)
    voxelize_fwd_kernel<4><<<grid_for(n * (c / 4), 256), 256, 0, st>>>(feats, idx, counts, n, c, m, out);
  else
    voxelize_fwd_kernel<1><<<grid_for(n * c, 256), 256, 0, st>>>(feats, idx, counts, n, c, m, out);
  return check_launch("ftx_voxelize_fwd");
}

template <int VEC>
__global__ void voxelize_bwd_kernel(const float *__restrict__ go, const int32_t *__restrict__ idx, const int32_t *__restrict__ counts,
                                    int64_t n, int c, int64_t m, float *__restrict__ gf) {
  const int cv = c / VEC;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * VEC;
    int32_t pos = idx[i];
    int32_t cnt = (pos >= 0 && pos < m) ? counts[pos] : 0;
    if (cnt > 0) {
      const float fc = (float)cnt;
#pragma unroll
      for (int v = 0; v < VEC; ++v) gf[i * c + j + v] = go[(int64_t)pos * c + j + v] / fc;
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) gf[i * c + j + v] = 0.f;
    }
  }
}

extern "C" int ftx_voxelize_bwd(const float *grad_out, const int32_t *idx, const int32_t *counts, int64_t n, int32_t c, int64_t m,
                                float *grad_feats, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 1, "ftx_voxelize_bwd: bad size");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(grad_out && idx && counts && grad_feats, "ftx_voxelize_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (c % 4 == 0)
    voxelize_bwd_kernel<4><<<grid_for(n * (c / 4), 256), 256, 0, st>>>(grad_out, idx, counts, n, c, m, grad_feats);
  else
    voxelize_bwd_kernel<1><<<grid_for(n * c, 256), 256, 0, st>>>(grad_out, idx, counts, n, c, m, grad_feats);
  return check_launch("ftx_voxelize_bwd");
}

// ---------------------------------------------------------------- devoxelize
// out[i, :] = sum_k w[i,k] * feats[idx[i,k], :]   (8 corner rows per point)
__global__ void devoxelize_fwd_kernel(const float *__restrict__ feats, const int32_t *__restrict__ idx, const float *__restrict__ w,
                                      int64_t n, int c, int64_t m, float *__restrict__ out) {
  const int cv = c / 4;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int32_t r = idx[i * 8 + k];
      if (r >= 0 && r < m) {
        float wk = w[i * 8 + k];
        float4 f = *(const float4 *)&feats[(int64_t)r * c + j];
        acc.x += wk * f.x; acc.y += wk * f.y; acc.z += wk * f.z; acc.w += wk * f.w;
      }
    }
    *(float4 *)&out[i * c + j] = acc;
  }
}

extern "C" int ftx_devoxelize_fwd(const float *feats, const int32_t *idx, const float *weights, int64_t n, int32_t c, int64_t m,
                                  float *out, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 4 && c % 4 == 0, "ftx_devoxelize_fwd: bad size (c must be a multiple of 4)");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(idx && weights && out && (feats || m == 0), "ftx_devoxelize_fwd: null pointer");
  devoxelize_fwd_kernel<<<grid_for(n * (c / 4), 256), 256, 0, (hipStream_t)stream>>>(feats, idx, weights, n, c, m, out);
  return check_launch("ftx_devoxelize_fwd");
}

__global__ void devoxelize_bwd_kernel(const float *__restrict__ go, const int32_t *__restrict__ idx, const float *__restrict__ w,
                                      int64_t n, int c, int64_t m, float *__restrict__ gf) {
  const int cv = c / 4;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * 4;
    float4 g = *(const float4 *)&go[i * c + j];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int32_t r = idx[i * 8 + k];
      float wk = w[i * 8 + k];
      if (r >= 0 && r < m && wk != 0.f) {
        float *dst = &gf[(int64_t)r * c + j];
        atomicAdd(dst + 0, wk * g.x);
        atomicAdd(dst + 1, wk * g.y);
        atomicAdd(dst + 2, wk * g.z);
        atomicAdd(dst + 3, wk * g.w);
      }
    }
  }
}

extern "C" int ftx_devoxelize_bwd(const float *grad_out, const int32_t *idx, const float *weights, int64_t n, int32_t c, int64_t m,
                                  float *grad_feats, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 4 && c % 4 == 0, "ftx_devoxelize_bwd: bad size (c must be a multiple of 4)");
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(grad_feats && ((grad_out && idx && weights) || n == 0), "ftx_devoxelize_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(grad_feats, 0, sizeof(float) * m * c, st) != hipSuccess) return check_launch("ftx_devoxelize_bwd memset");
  if (n == 0) return FTX_OK;
  devoxelize_bwd_kernel<<<grid_for(n * (c / 4), 256), 256, 0, st>>>(grad_out, idx, weights, n, c, m, grad_feats);
  return check_launch("ftx_devoxelize_bwd");
}

// ---------------------------------------------------------------- lift gather
// nn.Upsample(size) nearest source index in float32, exactly as ATen computes it:
// src = min((int)floorf(dst * ((float)n_in / (float)n_out)), n_in - 1).
__device__ inline int nearest_src(int dst, int n_in, int n_out) {
  float scale = (float)n_in / (float)n_out;
  int s = (int)floorf((float)dst * scale);
  return s < n_in - 1 ? s : n_in - 1;
}

__global__ void lift_gather_fwd_kernel(const float *__restrict__ grid, const int64_t *__restrict__ img_idx,
                                       const int32_t *__restrict__ pb, int64_t n, int gh, int gw, int c, int H, int W,
                                       float *__restrict__ out) {
  const int cv = c / 4;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * 4;
    int sr = nearest_src((int)img_idx[i * 2 + 0], gh, H);
    int sc = nearest_src((int)img_idx[i * 2 + 1], gw, W);
    int64_t cell = ((int64_t)pb[i] * gh + sr) * gw + sc;
    *(float4 *)&out[i * c + j] = *(const float4 *)&grid[cell * c + j];
  }
}

__global__ void lift_gather_bwd_kernel(const float *__restrict__ go, const int64_t *__restrict__ img_idx, const int32_t *__restrict__ pb,
                                       int64_t n, int gh, int gw, int c, int H, int W, float *__restrict__ gg) {
  const int cv = c / 4;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = e / cv;
    int j = (int)(e - i * cv) * 4;
    int sr = nearest_src((int)img_idx[i * 2 + 0], gh, H);
    int sc = nearest_src((int)img_idx[i * 2 + 1], gw, W);
    int64_t cell = ((int64_t)pb[i] * gh + sr) * gw + sc;
    float4 g = *(const float4 *)&go[i * c + j];
    float *dst = &gg[cell * c + j];
    atomicAdd(dst + 0, g.x);
    atomicAdd(dst + 1, g.y);
    atomicAdd(dst + 2, g.z);
    atomicAdd(dst + 3, g.w);
  }
}

static int lift_check(const char *who, int64_t n, int b, int gh, int gw, int c, int H, int W) {
  FTX_REQUIRE(n >= 0 && b >= 1 && gh >= 1 && gw >= 1 && H >= 1 && W >= 1, "%s: bad size", who);
  FTX_REQUIRE(c >= 4 && c % 4 == 0, "%s: c must be a multiple of 4", who);
  return FTX_OK;
}

extern "C" int ftx_lift_gather_fwd(const float *grid, const int64_t *img_idx, const int32_t *point_batch, int64_t n, int32_t b,
                                   int32_t gh, int32_t gw, int32_t c, int32_t H, int32_t W, float *out, void *stream) {
  int rc = lift_check("ftx_lift_gather_fwd", n, b, gh, gw, c, H, W);
  if (rc != FTX_OK) return rc;
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(grid && img_idx && point_batch && out, "ftx_lift_gather_fwd: null pointer");
  lift_gather_fwd_kernel<<<grid_for(n * (c / 4), 256), 256, 0, (hipStream_t)stream>>>(grid, img_idx, point_batch, n, gh, gw, c, H, W, out);
  return check_launch("ftx_lift_gather_fwd");
}

extern "C" int ftx_lift_gather_bwd(const float *grad_out, const int64_t *img_idx, const int32_t *point_batch, int64_t n, int32_t b,
                                   int32_t gh, int32_t gw, int32_t c, int32_t H, int32_t W, float *grad_grid, void *stream) {
  int rc = lift_check("ftx_lift_gather_bwd", n, b, gh, gw, c, H, W);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(grad_grid, "ftx_lift_gather_bwd: null grad_grid");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(grad_grid, 0, sizeof(float) * (size_t)b * gh * gw * c, st) != hipSuccess) return check_launch("ftx_lift_gather_bwd memset");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(grad_out && img_idx && point_batch, "ftx_lift_gather_bwd: null pointer");
  lift_gather_bwd_kernel<<<grid_for(n * (c / 4), 256), 256, 0, st>>>(grad_out, img_idx, point_batch, n, gh, gw, c, H, W, grad_grid);
  return check_launch("ftx_lift_gather_bwd");
}

// ---------------------------------------------------------------- NCHW nearest resample
__global__ void resample_fwd_kernel(const float *__restrict__ in, int64_t planes, int ih, int iw, int oh, int ow, float *__restrict__ out) {
  const int64_t total = planes * oh * ow;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int ox = (int)(e % ow);
    int64_t t = e / ow;
    int oy = (int)(t % oh);
    int64_t p = t / oh;
    out[e] = in[(p * ih + nearest_src(oy, ih, oh)) * iw + nearest_src(ox, iw, ow)];
  }
}

__global__ void resample_bwd_kernel(const float *__restrict__ go, int64_t planes, int ih, int iw, int oh, int ow, float *__restrict__ gi) {
  const int64_t total = planes * oh * ow;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int ox = (int)(e % ow);
    int64_t t = e / ow;
    int oy = (int)(t % oh);
    int64_t p = t / oh;
    atomicAdd(&gi[(p * ih + nearest_src(oy, ih, oh)) * iw + nearest_src(ox, iw, ow)], go[e]);
  }
}

extern "C" int ftx_resample_nearest_fwd(const float *in, int32_t b, int32_t c, int32_t ih, int32_t iw, int32_t oh, int32_t ow, float *out,
                                        void *stream) {
  FTX_REQUIRE(b >= 1 && c >= 1 && ih >= 1 && iw >= 1 && oh >= 1 && ow >= 1, "ftx_resample_nearest_fwd: bad size");
  FTX_REQUIRE(in && out, "ftx_resample_nearest_fwd: null pointer");
  int64_t planes = (int64_t)b * c;
  resample_fwd_kernel<<<grid_for(planes * oh * ow, 256), 256, 0, (hipStream_t)stream>>>(in, planes, ih, iw, oh, ow, out);
  return check_launch("ftx_resample_nearest_fwd");
}

extern "C" int ftx_resample_nearest_bwd(const float *grad_out, int32_t b, int32_t c, int32_t ih, int32_t iw, int32_t oh, int32_t ow,
                                        float *grad_in, void *stream) {
  FTX_REQUIRE(b >= 1 && c >= 1 && ih >= 1 && iw >= 1 && oh >= 1 && ow >= 1, "ftx_resample_nearest_bwd: bad size");
  FTX_REQUIRE(grad_out && grad_in, "ftx_resample_nearest_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int64_t planes = (int64_t)b * c;
  if (hipMemsetAsync(grad_in, 0, sizeof(float) * planes * ih * iw, st) != hipSuccess) return check_launch("ftx_resample_nearest_bwd memset");
  resample_bwd_kernel<<<grid_for(planes * oh * ow, 256), 256, 0, st>>>(grad_out, planes, ih, iw, oh, ow, grad_in);
  return check_launch("ftx_resample_nearest_bwd");
}
