// LayerNorm over the token rows of the ViT trunk, fused with the residual add in front of it
// (timm Block.forward as the reference runs it, models/transformers.py:16-45:
//   x = x + attn(norm1(x));  x = x + mlp(norm2(x))).
// forward:   s = x + y (y optional), h = (s - mean) * rstd * gamma + beta          one pass over the row
// backward:  gx = gs (optional) + rstd * (gy - mean(gy) - xhat * mean(gy * xhat)),  gy = gh * gamma,
//            d gamma = sum_rows gh * xhat,  d beta = sum_rows gh                     one pass + a column-sum pass
// y may come with the bias of the Linear that produced it (y_bias): the proj / fc2 GEMMs of a block then run without a bias epilogue,
// s = x + (y + b) in the epilogue's own rounding order, and the bias gradient (the column sums of gx) falls out of the backward pass
// as a third partial row instead of a column-sum launch pair per Linear.
// The add kernel, the LayerNorm kernel and (backward) the three LayerNorm-gradient kernels plus the residual-gradient add of the
// eager formulation become one launch forward and two backward.  HBM-bound: a row is 3 KB; one wave per row, the row lives in
// registers (C = 256 * VPL floats, VPL float4 per lane), statistics by wave shuffles, no LDS in the forward.
#include "ftx_common.h"

using namespace ftx;

// backward: rows per block (4 waves).  Few rows (batch 1: 578) => one row per wave, so that the launch still has ~150 blocks
static int ln_rows_per_block(int64_t rows) { return rows >= 2048 ? 16 : (rows >= 1024 ? 8 : 4); }

__device__ inline float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int VPL>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float *__restrict__ x, const float *__restrict__ y, const float *__restrict__ ybias,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int64_t rows,
                                                         float *__restrict__ s_out,
                                                         float *__restrict__ h_out, float *__restrict__ mean_out, float *__restrict__ rstd_out) {
  constexpr int C = 256 * VPL;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  float4 s[VPL];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int64_t o = row * C + (v * 64 + lane) * 4;
    s[v] = *(const float4 *)&x[o];
    if (y) {
      float4 t = *(const float4 *)&y[o];
      if (ybias) {   // y is a Linear's output without its bias: (y + b) first, as the GEMM epilogue would have rounded it
        const float4 bb = *(const float4 *)&ybias[(v * 64 + lane) * 4];
        t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w;
      }
      s[v].x += t.x; s[v].y += t.y; s[v].z += t.z; s[v].w += t.w;
      *(float4 *)&s_out[o] = s[v];
    }
    sum += (s[v].x + s[v].y) + (s[v].z + s[v].w);
  }
  const float mean = wave_sum_f(sum) * (1.f / C);
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const float a = s[v].x - mean, b = s[v].y - mean, c = s[v].z - mean, d = s[v].w - mean;
    sq += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum_f(sq) * (1.f / C) + eps);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int col = (v * 64 + lane) * 4;
    const float4 g = *(const float4 *)&gamma[col], b = *(const float4 *)&beta[col];
    float4 h;
    h.x = (s[v].x - mean) * rstd * g.x + b.x;
    h.y = (s[v].y - mean) * rstd * g.y + b.y;
    h.z = (s[v].z - mean) * rstd * g.z + b.z;
    h.w = (s[v].w - mean) * rstd * g.w + b.w;
    *(float4 *)&h_out[row * C + col] = h;
  }
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

template <int VPL, int NP>   // NP = 2: (d gamma, d beta); 3: also the column sums of gx (the gradient of y's bias)
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const float *__restrict__ gh, const float *__restrict__ gs, const float *__restrict__ s,
                                                         const float *__restrict__ gamma, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, int64_t rows, int rpb, float *__restrict__ gx,
                                                         double *__restrict__ part) {
  constexpr int C = 256 * VPL;
  __shared__ float sh[4][NP * C];   // the four waves' (d gamma, d beta[, d ybias]) rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 gm[VPL], dg[VPL], db[VPL], dyb[NP == 3 ? VPL : 1];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    gm[v] = *(const float4 *)&gamma[(v * 64 + lane) * 4];
    dg[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[v] = dg[v];
    if (NP == 3) dyb[v] = dg[v];
  }
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  for (int i = wave; i < rpb; i += 4) {   // this wave's rows, ascending
    const int64_t row = r0 + i;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float4 g[VPL], xh[VPL];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int64_t o = row * C + (v * 64 + lane) * 4;
      g[v] = *(const float4 *)&gh[o];
      const float4 sv = *(const float4 *)&s[o];
      xh[v].x = (sv.x - mu) * rs; xh[v].y = (sv.y - mu) * rs; xh[v].z = (sv.z - mu) * rs; xh[v].w = (sv.w - mu) * rs;
      dg[v].x += g[v].x * xh[v].x; dg[v].y += g[v].y * xh[v].y; dg[v].z += g[v].z * xh[v].z; dg[v].w += g[v].w * xh[v].w;
      db[v].x += g[v].x; db[v].y += g[v].y; db[v].z += g[v].z; db[v].w += g[v].w;
      g[v].x *= gm[v].x; g[v].y *= gm[v].y; g[v].z *= gm[v].z; g[v].w *= gm[v].w;   // gy = gh * gamma
      c1 += (g[v].x + g[v].y) + (g[v].z + g[v].w);
      c2 += (g[v].x * xh[v].x + g[v].y * xh[v].y) + (g[v].z * xh[v].z + g[v].w * xh[v].w);
    }
    c1 = wave_sum_f(c1) * (1.f / C);
    c2 = wave_sum_f(c2) * (1.f / C);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int64_t o = row * C + (v * 64 + lane) * 4;
      float4 d;
      d.x = rs * (g[v].x - c1 - xh[v].x * c2);
      d.y = rs * (g[v].y - c1 - xh[v].y * c2);
      d.z = rs * (g[v].z - c1 - xh[v].z * c2);
      d.w = rs * (g[v].w - c1 - xh[v].w * c2);
      if (gs) {
        const float4 t = *(const float4 *)&gs[o];
        d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
      }
      *(float4 *)&gx[o] = d;
      if (NP == 3) { dyb[v].x += d.x; dyb[v].y += d.y; dyb[v].z += d.z; dyb[v].w += d.w; }
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int col = (v * 64 + lane) * 4;
    *(float4 *)&sh[wave][col] = dg[v];
    *(float4 *)&sh[wave][C + col] = db[v];
    if (NP == 3) *(float4 *)&sh[wave][2 * C + col] = dyb[v];
  }
  __syncthreads();
  for (int j = threadIdx.x; j < NP * C; j += 256)
    part[(int64_t)blockIdx.x * (NP * C) + j] = ((double)sh[0][j] + (double)sh[1][j]) + ((double)sh[2][j] + (double)sh[3][j]);
}

// out[c] = sum over the chunk rows of part[k][c], 16 lanes per column (every 16th row each, combined in lane order)
__global__ __launch_bounds__(256) void ln_params_final_kernel(const double *__restrict__ part, int chunks, int cols, float *__restrict__ out) {
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

static int ln_check(const char *who, int64_t rows, int c) {
  FTX_REQUIRE(rows >= 0, "%s: rows < 0", who);
  FTX_REQUIRE(c >= 256 && c <= 1024 && c % 256 == 0, "%s: the row length must be 256, 512, 768 or 1024 (got %d)", who, c);
  return FTX_OK;
}

extern "C" int ftx_add_layernorm_fwd(const float *x, const float *y, const float *y_bias, const float *gamma, const float *beta, float eps,
                                     int64_t rows, int32_t c, float *s_out, float *h_out, float *mean, float *rstd, void *stream) {
  int rc = ln_check("ftx_add_layernorm_fwd", rows, c);
  if (rc != FTX_OK) return rc;
  if (rows == 0) return FTX_OK;
  FTX_REQUIRE(x && gamma && beta && h_out && mean && rstd, "ftx_add_layernorm_fwd: null pointer");
  FTX_REQUIRE(!y || s_out, "ftx_add_layernorm_fwd: the sum x + y needs an output");
  FTX_REQUIRE(!y_bias || y, "ftx_add_layernorm_fwd: y_bias without y");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)ceil_div(rows, 4);
  switch (c / 256) {
    case 1: add_ln_fwd_kernel<1><<<grid, 256, 0, st>>>(x, y, y_bias, gamma, beta, eps, rows, s_out, h_out, mean, rstd); break;
    case 2: add_ln_fwd_kernel<2><<<grid, 256, 0, st>>>(x, y, y_bias, gamma, beta, eps, rows, s_out, h_out, mean, rstd); break;
    case 3: add_ln_fwd_kernel<3><<<grid, 256, 0, st>>>(x, y, y_bias, gamma, beta, eps, rows, s_out, h_out, mean, rstd); break;
    default: add_ln_fwd_kernel<4><<<grid, 256, 0, st>>>(x, y, y_bias, gamma, beta, eps, rows, s_out, h_out, mean, rstd); break;
  }
  return check_launch("ftx_add_layernorm_fwd");
}

extern "C" size_t ftx_layernorm_bwd_workspace_bytes(int64_t rows, int32_t c) {
  if (rows <= 0 || c <= 0) return 256;
  return sizeof(double) * (size_t)ceil_div(rows, ln_rows_per_block(rows)) * 3 * (size_t)c + 256;
}

extern "C" int ftx_add_layernorm_bwd(const float *grad_h, const float *grad_s, const float *s, const float *gamma, const float *mean,
                                     const float *rstd, int64_t rows, int32_t c, int32_t with_y_bias, float *grad_x, float *grad_params,
                                     void *workspace, size_t workspace_bytes, void *stream) {
  int rc = ln_check("ftx_add_layernorm_bwd", rows, c);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(rows >= 1, "ftx_add_layernorm_bwd: needs at least one row");
  FTX_REQUIRE(grad_h && s && gamma && mean && rstd && grad_x && grad_params && workspace, "ftx_add_layernorm_bwd: null pointer");
  if (workspace_bytes < ftx_layernorm_bwd_workspace_bytes(rows, c)) {
    set_error("ftx_add_layernorm_bwd: workspace %zu < required %zu", workspace_bytes, ftx_layernorm_bwd_workspace_bytes(rows, c));
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int rpb = ln_rows_per_block(rows);
  const int nb = (int)ceil_div(rows, rpb);
  double *part = (double *)workspace;
  const int np = with_y_bias ? 3 : 2;
#define FTX_LN_BWD(VPL)                                                                                                             \
  do {                                                                                                                              \
    if (np == 3) add_ln_bwd_kernel<VPL, 3><<<nb, 256, 0, st>>>(grad_h, grad_s, s, gamma, mean, rstd, rows, rpb, grad_x, part);      \
    else add_ln_bwd_kernel<VPL, 2><<<nb, 256, 0, st>>>(grad_h, grad_s, s, gamma, mean, rstd, rows, rpb, grad_x, part);              \
  } while (0)
  switch (c / 256) {
    case 1: FTX_LN_BWD(1); break;
    case 2: FTX_LN_BWD(2); break;
    case 3: FTX_LN_BWD(3); break;
    default: FTX_LN_BWD(4); break;
  }
#undef FTX_LN_BWD
  ln_params_final_kernel<<<(unsigned)ceil_div(np * c, 16), 256, 0, st>>>(part, nb, np * c, grad_params);
  return check_launch("ftx_add_layernorm_bwd");
}
