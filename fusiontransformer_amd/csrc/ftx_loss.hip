// Fused segmentation losses + metric of the train step (modules/SemanticTrainer.py:158-194,
// models/metric.py:37-58): weighted cross-entropy on the 3-D and 2-D main heads, cross-modal KL terms
// on the second heads, the gradients of (loss_2d + loss_3d) with respect to all four logit tensors and
// both SegIoU confusion matrices, in one pass over the points: ~40 small framework launches and the
// host synchronisations of the boolean-mask indexing become 3 launches and none.
#include "ftx_common.h"

using namespace ftx;

constexpr int LC_MAX = 32;      // classes held in registers
constexpr int LOSS_BLOCKS = 256;
constexpr int WSUM_BLOCKS = 64;

__device__ inline double loss_block_sum(double v, double *sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
  return r;
}

// wpart[b] = sum of w[label_i] over block b's contiguous slice of the points (the normaliser of the weighted-mean cross-entropy is
// the sum of the WSUM_BLOCKS slices in slice order: loss_wsum_total).  Depends on the labels only.  One block of 1024 threads took
// 71 us for 81 k points -- on the critical path between the forward and the backward --, 64 blocks take 5.
__global__ __launch_bounds__(256) void loss_wsum_kernel(const int64_t *__restrict__ label, const float *__restrict__ cw, int64_t n, int c,
                                                        double *__restrict__ wpart) {
  __shared__ double sh[4];
  const int64_t per = ceil_div(n, (int64_t)gridDim.x);
  const int64_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double s = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int64_t y = label[i];
    if (y >= 0 && y < c) s += cw ? (double)cw[y] : 1.0;
  }
  double r = loss_block_sum(s, sh);
  if (threadIdx.x == 0) wpart[blockIdx.x] = r;
}
__device__ inline double loss_wsum_total(const double *__restrict__ wpart) {
  double W = 0;
  for (int b = 0; b < WSUM_BLOCKS; ++b) W += wpart[b];
  return W;
}

// One row of C logits in registers.  Every index into v[] is a compile-time constant (a row indexed by the label went to scratch memory
// and, with eleven such rows alive, the kernel held 512 registers and spilled 135: 100 us for 81 k points; now 3 rows are alive).
template <int C>
struct Row {
  float v[C];
};
template <int C>
__device__ inline void load_row(const float *p, Row<C> &r) {
#pragma unroll
  for (int j = 0; j < C; j += 4) {
    float4 t = *(const float4 *)(p + j);
    r.v[j] = t.x; r.v[j + 1] = t.y; r.v[j + 2] = t.z; r.v[j + 3] = t.w;
  }
}
template <int C>
__device__ inline void store_row(float *p, const Row<C> &r) {
#pragma unroll
  for (int j = 0; j < C; j += 4) *(float4 *)(p + j) = make_float4(r.v[j], r.v[j + 1], r.v[j + 2], r.v[j + 3]);
}
// x -> log_softmax(x) in place; returns the argmax (first maximum)
template <int C>
__device__ inline int log_softmax_row(Row<C> &r) {
  float m = -INFINITY;
  int amax = 0;
#pragma unroll
  for (int j = 0; j < C; ++j)
    if (r.v[j] > m) { m = r.v[j]; amax = j; }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < C; ++j) s += expf(r.v[j] - m);
  const float lse = logf(s) + m;
#pragma unroll
  for (int j = 0; j < C; ++j) r.v[j] -= lse;
  return amax;
}
template <int C>
__device__ inline float pick(const Row<C> &r, int y) {   // r.v[y] without a dynamic register index
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < C; ++j) t = (j == y) ? r.v[j] : t;
  return t;
}

// part[block][4] = { sum w*nll_3d, sum w*nll_2d, sum kl_2d, sum kl_3d }
template <int C>
__global__ __launch_bounds__(256) void loss_main_kernel(const float *__restrict__ l3, const float *__restrict__ l2, const float *__restrict__ l3b,
                                                        const float *__restrict__ l2b, const int64_t *__restrict__ label,
                                                        const float *__restrict__ cw, const double *__restrict__ wpart, float ce_scale, float lambda_xm,
                                                        int64_t n, int ignore_index, float *__restrict__ g3, float *__restrict__ g2,
                                                        float *__restrict__ g3b, float *__restrict__ g2b, long long *__restrict__ conf3,
                                                        long long *__restrict__ conf2, double *__restrict__ part) {
  constexpr int c = C;
  __shared__ double sh[4];
  __shared__ float s_invW;
  // confusion counts of this block, flushed once at the end: the points of a frame fall on a few dominant (label, prediction) cells,
  // and one global 64-bit atomic per point on those cells serialised the kernel (100 us for 22 k points)
  __shared__ unsigned int cf3[C * C], cf2[C * C];
  for (int j = threadIdx.x; j < c * c; j += blockDim.x) cf3[j] = cf2[j] = 0u;
  if (threadIdx.x == 0) s_invW = ce_scale * (float)(1.0 / loss_wsum_total(wpart));   // d(ce_scale * CE) / d(logits) carries the mix factor
  __syncthreads();
  const bool dual = (l3b != l3);
  const float invW = s_invW;
  const float invN = 1.f / (float)n;
  double a_ce3 = 0, a_ce2 = 0, a_kl2 = 0, a_kl3 = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    Row<C> lp3, lp2;     // log-probabilities of the two main heads; the probabilities are exp() of them where needed
    load_row<C>(l3 + i * c, lp3);
    load_row<C>(l2 + i * c, lp2);
    const int am3 = log_softmax_row<C>(lp3), am2 = log_softmax_row<C>(lp2);
    const int64_t y64 = label[i];
    const bool yv = y64 >= 0 && y64 < c;
    const int y = yv ? (int)y64 : -1;
    const float w = yv ? (cw ? cw[y] : 1.f) : 0.f;
    if (yv) {
      a_ce3 += (double)(-w * pick<C>(lp3, y));
      a_ce2 += (double)(-w * pick<C>(lp2, y));
      if (y != ignore_index) {
        if (conf3) atomicAdd(&cf3[y * c + am3], 1u);
        if (conf2) atomicAdd(&cf2[y * c + am2], 1u);
      }
    }
    const float wW = w * invW;
    const float s = lambda_xm * invN;
    Row<C> g;
    if (lambda_xm > 0.f && dual) {
      // KL(softmax(other main head) || softmax(this second head)), mean over points; one second head at a time
      Row<C> lq;
      load_row<C>(l2b + i * c, lq);
      log_softmax_row<C>(lq);
      float kl2 = 0.f;
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const float t3 = expf(lp3.v[j]);
        kl2 += t3 > 0.f ? t3 * (lp3.v[j] - lq.v[j]) : 0.f;   // target = softmax(lidar main), input = log_softmax(img second)
        g.v[j] = s * (expf(lq.v[j]) - t3);
      }
      a_kl2 += kl2;
      store_row<C>(g2b + i * c, g);
      load_row<C>(l3b + i * c, lq);
      log_softmax_row<C>(lq);
      float kl3 = 0.f;
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const float t2 = expf(lp2.v[j]);
        kl3 += t2 > 0.f ? t2 * (lp2.v[j] - lq.v[j]) : 0.f;
        g.v[j] = s * (expf(lq.v[j]) - t2);
      }
      a_kl3 += kl3;
      store_row<C>(g3b + i * c, g);
    } else if (dual) {
#pragma unroll
      for (int j = 0; j < C; ++j) g.v[j] = 0.f;
      store_row<C>(g2b + i * c, g);
      store_row<C>(g3b + i * c, g);
    }
    // cross-entropy gradients (weighted mean): w/W * (softmax - onehot); single head: the KL terms land on the same logits
    const bool kl_same = lambda_xm > 0.f && !dual;
    float kl2s = 0.f, kl3s = 0.f;
    Row<C> h;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float p3 = expf(lp3.v[j]), p2 = expf(lp2.v[j]);
      g.v[j] = wW * (p3 - ((j == y) ? 1.f : 0.f));
      h.v[j] = wW * (p2 - ((j == y) ? 1.f : 0.f));
      if (kl_same) {
        kl2s += p3 > 0.f ? p3 * (lp3.v[j] - lp2.v[j]) : 0.f;
        kl3s += p2 > 0.f ? p2 * (lp2.v[j] - lp3.v[j]) : 0.f;
        h.v[j] += s * (p2 - p3);
        g.v[j] += s * (p3 - p2);
      }
    }
    if (kl_same) {
      a_kl2 += kl2s;
      a_kl3 += kl3s;
    }
    store_row<C>(g3 + i * c, g);
    store_row<C>(g2 + i * c, h);
  }
  double r0 = loss_block_sum(a_ce3, sh), r1 = loss_block_sum(a_ce2, sh), r2 = loss_block_sum(a_kl2, sh), r3 = loss_block_sum(a_kl3, sh);
  if (threadIdx.x == 0) {
    double *p = part + (int64_t)blockIdx.x * 4;
    p[0] = r0; p[1] = r1; p[2] = r2; p[3] = r3;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < c * c; j += blockDim.x) {   // integer counts: the order of the adds does not matter
    if (conf3 && cf3[j]) atomicAdd((unsigned long long *)&conf3[j], (unsigned long long)cf3[j]);
    if (conf2 && cf2[j]) atomicAdd((unsigned long long *)&conf2[j], (unsigned long long)cf2[j]);
  }
}

// The four sums over the (at most 256) block rows by one block of 256 threads: thread b holds row b, the rows are added by a fixed
// shuffle tree.  (One thread walking the rows took 35 us.)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double *__restrict__ part, int nb, const double *__restrict__ wpart, float ce_scale,
                                                            float lambda_xm, int64_t n, float *__restrict__ losses) {
  __shared__ double sh[4];
  const int b = threadIdx.x;
  double v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = b < nb ? part[(int64_t)b * 4 + j] : 0.0;
  double s[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) s[j] = loss_block_sum(v[j], sh);
  if (threadIdx.x != 0) return;
  const double W = loss_wsum_total(wpart);
  const double ce3 = s[0] / W, ce2 = s[1] / W, kl2 = s[2] / (double)n, kl3 = s[3] / (double)n;
  losses[0] = (float)(ce_scale * ce2 + lambda_xm * kl2);   // loss_2d
  losses[1] = (float)(ce_scale * ce3 + lambda_xm * kl3);   // loss_3d
}

extern "C" size_t ftx_fusion_loss_workspace_bytes(void) { return sizeof(double) * (LOSS_BLOCKS * 4 + WSUM_BLOCKS) + 256; }

extern "C" int ftx_fusion_loss_mix(const float *lidar_logit, const float *img_logit, const float *lidar_logit2, const float *img_logit2,
                               const int64_t *label, const float *class_weights, float ce_scale, float lambda_xm, int64_t n, int32_t c, int32_t ignore_index,
                               float *losses, float *grad_lidar, float *grad_img, float *grad_lidar2, float *grad_img2, int64_t *conf3d,
                               int64_t *conf2d, void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n >= 1, "ftx_fusion_loss_mix: needs at least one point");
  FTX_REQUIRE(c >= 4 && c % 4 == 0 && c <= LC_MAX, "ftx_fusion_loss_mix: classes must be a multiple of 4 and <= %d (got %d)", LC_MAX, c);
  FTX_REQUIRE(lidar_logit && img_logit && label && losses && grad_lidar && grad_img && workspace, "ftx_fusion_loss_mix: null pointer");
  const bool dual = lidar_logit2 != nullptr || img_logit2 != nullptr;
  FTX_REQUIRE(!dual || (lidar_logit2 && img_logit2 && grad_lidar2 && grad_img2), "ftx_fusion_loss_mix: dual head needs both second heads and their gradients");
  if (workspace_bytes < ftx_fusion_loss_workspace_bytes()) {
    set_error("ftx_fusion_loss_mix: workspace too small");
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double *part = (double *)workspace;
  double *wpart = part + LOSS_BLOCKS * 4;
  loss_wsum_kernel<<<WSUM_BLOCKS, 256, 0, st>>>(label, class_weights, n, c, wpart);
  const int nb = (int)(ceil_div(n, 256) < LOSS_BLOCKS ? ceil_div(n, 256) : LOSS_BLOCKS);
#define LOSS_CASE(C_)                                                                                                                            \
  case C_:                                                                                                                                       \
    loss_main_kernel<C_><<<nb, 256, 0, st>>>(lidar_logit, img_logit, dual ? lidar_logit2 : lidar_logit, dual ? img_logit2 : img_logit, label,   \
                                             class_weights, wpart, ce_scale, lambda_xm, n, ignore_index, grad_lidar, grad_img,                  \
                                             dual ? grad_lidar2 : grad_lidar, dual ? grad_img2 : grad_img, (long long *)conf3d,                 \
                                             (long long *)conf2d, part);                                                                        \
    break
  switch (c) {
    LOSS_CASE(4); LOSS_CASE(8); LOSS_CASE(12); LOSS_CASE(16); LOSS_CASE(20); LOSS_CASE(24); LOSS_CASE(28); LOSS_CASE(32);
  }
#undef LOSS_CASE
  loss_finalize_kernel<<<1, 256, 0, st>>>(part, nb, wpart, ce_scale, lambda_xm, n, losses);
  return check_launch("ftx_fusion_loss_mix");
}

// The additive mix of modules/SemanticTrainer.py:158-178: CE + lambda * KL.
extern "C" int ftx_fusion_loss(const float *lidar_logit, const float *img_logit, const float *lidar_logit2, const float *img_logit2,
                               const int64_t *label, const float *class_weights, float lambda_xm, int64_t n, int32_t c, int32_t ignore_index,
                               float *losses, float *grad_lidar, float *grad_img, float *grad_lidar2, float *grad_img2, int64_t *conf3d,
                               int64_t *conf2d, void *workspace, size_t workspace_bytes, void *stream) {
  return ftx_fusion_loss_mix(lidar_logit, img_logit, lidar_logit2, img_logit2, label, class_weights, 1.f, lambda_xm, n, c, ignore_index, losses,
                             grad_lidar, grad_img, grad_lidar2, grad_img2, conf3d, conf2d, workspace, workspace_bytes, stream);
}
