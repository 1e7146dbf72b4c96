// Point <-> voxel feature movement (spvoxelize / spdevoxelize fwd+bwd), the fused
// nearest-upsample + per-point lift gather, and the NCHW nearest resample.
// HBM-bound row gathers/scatters: one thread moves 16 bytes of one row, lanes of a
// wave cover consecutive 16-byte pieces of the same row (coalesced 128..1024 B).
#include <cstring>
#include <cstdlib>
#include "ftx_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

// ---------------------------------------------------------------- sorted segments (no float atomics)
// The scatter sides of voxelize (forward) and devoxelize (backward) as gather-reduces: entries
// are sorted by destination row once per (batch, stride); each destination then sums its own
// entries in ascending entry order -> plain loads/stores at stream rate instead of the ~1.3 TB/s
// float-atomic rate, and bit-reproducible.
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct SegLayout {
  size_t off_keys_in, off_keys_out, off_vals_in, off_tmp, tmp_bytes, total;
};

static int seg_layout(int64_t n, int64_t m, SegLayout *L) {
  size_t sort_bytes = 0, scan_bytes = 0;
  int32_t *kp = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, kp, kp, (size_t)n, 0u, 32u) != hipSuccess ||
      rocprim::exclusive_scan(nullptr, scan_bytes, kp, kp, 0, (size_t)(m + 1), rocprim::plus<int32_t>()) != hipSuccess) {
    set_error("ftx_segment_build: rocprim size query failed");
    return FTX_ELAUNCH;
  }
  L->off_keys_in = 0;
  L->off_keys_out = align256(sizeof(int32_t) * n);
  L->off_vals_in = align256(L->off_keys_out + sizeof(int32_t) * n);
  L->off_tmp = align256(L->off_vals_in + sizeof(int32_t) * n);
  L->tmp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  L->total = align256(L->off_tmp + L->tmp_bytes);
  return FTX_OK;
}

extern "C" size_t ftx_segment_workspace_bytes(int64_t n, int64_t m) {
  if (n <= 0 || m < 0) return 256;
  SegLayout L;
  if (seg_layout(n, m, &L) != FTX_OK) return 0;
  return L.total;
}

__global__ void seg_prepare_kernel(const int32_t *__restrict__ keys, int64_t n, int64_t m, int32_t *__restrict__ keys_in,
                                   int32_t *__restrict__ vals_in, int32_t *__restrict__ counts) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t k = keys[i];
    bool ok = k >= 0 && k < m;
    keys_in[i] = ok ? k : (int32_t)m;   // invalid entries sort to the end
    vals_in[i] = (int32_t)i;
    if (ok) atomicAdd(&counts[k], 1);
  }
}

extern "C" int ftx_segment_build(const int32_t *keys, int64_t n, int64_t m, int32_t *order, int32_t *seg_off, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && n < 0x7fffffff && m < 0x7ffffffe, "ftx_segment_build: bad size");
  FTX_REQUIRE(seg_off, "ftx_segment_build: null seg_off");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(seg_off, 0, sizeof(int32_t) * (m + 1), st) != hipSuccess) return check_launch("ftx_segment_build memset");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(keys && order && workspace, "ftx_segment_build: null pointer");
  SegLayout L;
  int rc = seg_layout(n, m, &L);
  if (rc != FTX_OK) return rc;
  if (workspace_bytes < L.total) {
    set_error("ftx_segment_build: workspace %zu < required %zu", workspace_bytes, L.total);
    return FTX_EWORKSPACE;
  }
  char *ws = (char *)workspace;
  int32_t *keys_in = (int32_t *)(ws + L.off_keys_in), *keys_out = (int32_t *)(ws + L.off_keys_out), *vals_in = (int32_t *)(ws + L.off_vals_in);
  void *tmp = ws + L.off_tmp;
  seg_prepare_kernel<<<grid_for(n, 256), 256, 0, st>>>(keys, n, m, keys_in, vals_in, seg_off);  // seg_off holds the counts for now
  size_t tb = L.tmp_bytes;
  unsigned bits = 1;
  while ((1ll << bits) <= m) ++bits;
  if (rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, vals_in, order, (size_t)n, 0u, bits, st) != hipSuccess) {
    set_error("ftx_segment_build: sort failed");
    return FTX_ELAUNCH;
  }
  tb = L.tmp_bytes;
  if (rocprim::exclusive_scan(tmp, tb, seg_off, seg_off, 0, (size_t)(m + 1), rocprim::plus<int32_t>(), st) != hipSuccess) {
    set_error("ftx_segment_build: scan failed");
    return FTX_ELAUNCH;
  }
  return check_launch("ftx_segment_build");
}

// out[v,:] = sum over the entries e of segment v of  scale(e) * src[row(e),:]
//   voxelize fwd:    row(e) = e,      scale = 1 / (segment length)
//   devoxelize bwd:  row(e) = e >> 3, scale = w[e]
// MODE 0: mean of src[e] (voxelize fwd); 1: sum of w[e] * src[e >> 3] (devoxelize bwd); 2: plain sum of src[e]
template <int MODE>
__global__ void segment_reduce_kernel(const float *__restrict__ src, const float *__restrict__ w, const int32_t *__restrict__ order,
                                      const int32_t *__restrict__ seg_off, int64_t m, int c, float *__restrict__ out) {
  const int cv = c >> 2;
  const int64_t total = m * cv;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = t / cv;
    int j = (int)(t - v * cv) * 4;
    const int lo = seg_off[v], hi = seg_off[v + 1];
    constexpr bool DEVOX = MODE == 1;
    const float inv = MODE == 0 ? (float)(hi - lo) : 1.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = lo;
    for (; e + 4 <= hi; e += 4) {   // four independent row loads in flight
      int32_t o0 = order[e], o1 = order[e + 1], o2 = order[e + 2], o3 = order[e + 3];
      float4 f0 = *(const float4 *)&src[(int64_t)(DEVOX ? o0 >> 3 : o0) * c + j];
      float4 f1 = *(const float4 *)&src[(int64_t)(DEVOX ? o1 >> 3 : o1) * c + j];
      float4 f2 = *(const float4 *)&src[(int64_t)(DEVOX ? o2 >> 3 : o2) * c + j];
      float4 f3 = *(const float4 *)&src[(int64_t)(DEVOX ? o3 >> 3 : o3) * c + j];
      if (DEVOX) {
        float w0 = w[o0], w1 = w[o1], w2 = w[o2], w3 = w[o3];
        acc.x += w0 * f0.x; acc.y += w0 * f0.y; acc.z += w0 * f0.z; acc.w += w0 * f0.w;
        acc.x += w1 * f1.x; acc.y += w1 * f1.y; acc.z += w1 * f1.z; acc.w += w1 * f1.w;
        acc.x += w2 * f2.x; acc.y += w2 * f2.y; acc.z += w2 * f2.z; acc.w += w2 * f2.w;
        acc.x += w3 * f3.x; acc.y += w3 * f3.y; acc.z += w3 * f3.z; acc.w += w3 * f3.w;
      } else {
        acc.x += f0.x / inv; acc.y += f0.y / inv; acc.z += f0.z / inv; acc.w += f0.w / inv;
        acc.x += f1.x / inv; acc.y += f1.y / inv; acc.z += f1.z / inv; acc.w += f1.w / inv;
        acc.x += f2.x / inv; acc.y += f2.y / inv; acc.z += f2.z / inv; acc.w += f2.w / inv;
        acc.x += f3.x / inv; acc.y += f3.y / inv; acc.z += f3.z / inv; acc.w += f3.w / inv;
      }
    }
    for (; e < hi; ++e) {
      int32_t o = order[e];
      float4 f = *(const float4 *)&src[(int64_t)(DEVOX ? o >> 3 : o) * c + j];
      if (DEVOX) {
        float wk = w[o];
        acc.x += wk * f.x; acc.y += wk * f.y; acc.z += wk * f.z; acc.w += wk * f.w;
      } else {
        acc.x += f.x / inv; acc.y += f.y / inv; acc.z += f.z / inv; acc.w += f.w / inv;
      }
    }
    *(float4 *)&out[v * c + j] = acc;
  }
}

extern "C" int ftx_voxelize_fwd_sorted(const float *feats, const int32_t *order, const int32_t *seg_off, int64_t n, int32_t c, int64_t m,
                                       float *out, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 4 && c % 4 == 0, "ftx_voxelize_fwd_sorted: bad size (c must be a multiple of 4)");
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(seg_off && out && ((feats && order) || n == 0), "ftx_voxelize_fwd_sorted: null pointer");
  segment_reduce_kernel<0><<<grid_for(m * (c / 4), 256), 256, 0, (hipStream_t)stream>>>(feats, nullptr, order, seg_off, m, c, out);
  return check_launch("ftx_voxelize_fwd_sorted");
}

extern "C" int ftx_devoxelize_bwd_sorted(const float *grad_out, const float *weights, const int32_t *order, const int32_t *seg_off, int64_t n,
                                         int32_t c, int64_t m, float *grad_feats, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 4 && c % 4 == 0, "ftx_devoxelize_bwd_sorted: bad size (c must be a multiple of 4)");
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(seg_off && grad_feats && ((grad_out && weights && order) || n == 0), "ftx_devoxelize_bwd_sorted: null pointer");
  segment_reduce_kernel<1><<<grid_for(m * (c / 4), 256), 256, 0, (hipStream_t)stream>>>(grad_out, weights, order, seg_off, m, c, grad_feats);
  return check_launch("ftx_devoxelize_bwd_sorted");
}

// out[v,:] = sum of src[e,:] over the entries e of segment v (fixed order): the generic atomic-free scatter-add.
extern "C" int ftx_segment_sum(const float *src, const int32_t *order, const int32_t *seg_off, int64_t n, int32_t c, int64_t m, float *out,
                               void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && c >= 4 && c % 4 == 0, "ftx_segment_sum: bad size (c must be a multiple of 4)");
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(seg_off && out && ((src && order) || n == 0), "ftx_segment_sum: null pointer");
  segment_reduce_kernel<2><<<grid_for(m * (c / 4), 256), 256, 0, (hipStream_t)stream>>>(src, nullptr, order, seg_off, m, c, out);
  return check_launch("ftx_segment_sum");
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

__global__ void lift_cells_kernel(const int64_t *__restrict__ img_idx, const int32_t *__restrict__ pb, int64_t n, int b, int gh, int gw,
                                  int H, int W, int32_t *__restrict__ cells) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int sr = nearest_src((int)img_idx[i * 2 + 0], gh, H);
    int sc = nearest_src((int)img_idx[i * 2 + 1], gw, W);
    int fb = pb[i];
    cells[i] = (fb >= 0 && fb < b && sr >= 0 && sc >= 0) ? (fb * gh + sr) * gw + sc : -1;
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

// cells[i] = flat (frame, source row, source col) cell of point i in the (b, gh, gw) grid: the destination of the
// lift gather's backward, so that it can run as ftx_segment_build + ftx_segment_sum instead of float atomics.
extern "C" int ftx_lift_cells(const int64_t *img_idx, const int32_t *point_batch, int64_t n, int32_t b, int32_t gh, int32_t gw, int32_t H,
                              int32_t W, int32_t *cells, void *stream) {
  FTX_REQUIRE(n >= 0 && b >= 1 && gh >= 1 && gw >= 1 && H >= 1 && W >= 1, "ftx_lift_cells: bad size");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(img_idx && point_batch && cells, "ftx_lift_cells: null pointer");
  lift_cells_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(img_idx, point_batch, n, b, gh, gw, H, W, cells);
  return check_launch("ftx_lift_cells");
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
