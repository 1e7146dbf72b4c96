// Fused segmentation losses + metric of the train step (modules/SemanticTrainer.py:158-194,
// models/metric.py:37-58): weighted cross-entropy on the 3-D and 2-D main heads, cross-modal KL terms
// on the second heads, the gradients of (loss_2d + loss_3d) with respect to all four logit tensors and
// both SegIoU confusion matrices, in one pass over the points: ~40 small framework launches and the
// host synchronisations of the boolean-mask indexing become 3 launches and none.
#include "ftx_common.h"

using namespace ftx;

constexpr int LC_MAX = 32;      // classes held in registers
constexpr int LOSS_BLOCKS = 256;

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

// sum_i w[label_i]  (the normaliser of the weighted-mean cross-entropy); one block
__global__ __launch_bounds__(1024) void loss_wsum_kernel(const int64_t *__restrict__ label, const float *__restrict__ cw, int64_t n, int c,
                                                         double *__restrict__ wsum) {
  __shared__ double sh[16];
  double s = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    int64_t y = label[i];
    if (y >= 0 && y < c) s += cw ? (double)cw[y] : 1.0;
  }
  double r = loss_block_sum(s, sh);
  if (threadIdx.x == 0) wsum[0] = r;
}

struct Row {
  float v[LC_MAX];
};

__device__ inline void load_row(const float *p, int c, Row &r) {
#pragma unroll
  for (int j = 0; j < LC_MAX; j += 4)
    if (j < c) {
      float4 t = *(const float4 *)(p + j);
      r.v[j] = t.x; r.v[j + 1] = t.y; r.v[j + 2] = t.z; r.v[j + 3] = t.w;
    }
}
// softmax in place; returns log(sum exp(x - max)) + max  (so log_softmax_j = x_j - lse)
__device__ inline float softmax_row(Row &r, int c, Row &logp, int &amax) {
  float m = -INFINITY;
  amax = 0;
#pragma unroll
  for (int j = 0; j < LC_MAX; ++j)
    if (j < c && r.v[j] > m) { m = r.v[j]; amax = j; }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LC_MAX; ++j)
    if (j < c) s += expf(r.v[j] - m);
  const float lse = logf(s) + m;
#pragma unroll
  for (int j = 0; j < LC_MAX; ++j)
    if (j < c) {
      logp.v[j] = r.v[j] - lse;
      r.v[j] = expf(logp.v[j]);
    }
  return lse;
}
__device__ inline void store_row(float *p, int c, const Row &r) {
#pragma unroll
  for (int j = 0; j < LC_MAX; j += 4)
    if (j < c) *(float4 *)(p + j) = make_float4(r.v[j], r.v[j + 1], r.v[j + 2], r.v[j + 3]);
}

// part[block][4] = { sum w*nll_3d, sum w*nll_2d, sum kl_2d, sum kl_3d }
__global__ __launch_bounds__(256) void loss_main_kernel(const float *__restrict__ l3, const float *__restrict__ l2, const float *__restrict__ l3b,
                                                        const float *__restrict__ l2b, const int64_t *__restrict__ label,
                                                        const float *__restrict__ cw, const double *__restrict__ wsum, float ce_scale, float lambda_xm,
                                                        int64_t n, int c, int ignore_index, float *__restrict__ g3, float *__restrict__ g2,
                                                        float *__restrict__ g3b, float *__restrict__ g2b, long long *__restrict__ conf3,
                                                        long long *__restrict__ conf2, double *__restrict__ part) {
  __shared__ double sh[4];
  // confusion counts of this block, flushed once at the end: the points of a frame fall on a few dominant (label, prediction) cells,
  // and one global 64-bit atomic per point on those cells serialised the kernel (100 us for 22 k points)
  __shared__ unsigned int cf3[LC_MAX * LC_MAX], cf2[LC_MAX * LC_MAX];
  for (int j = threadIdx.x; j < c * c; j += blockDim.x) cf3[j] = cf2[j] = 0u;
  __syncthreads();
  const bool dual = (l3b != l3);
  const float invW = ce_scale * (float)(1.0 / wsum[0]);   // d(ce_scale * CE) / d(logits) carries the mix factor
  const float invN = 1.f / (float)n;
  double a_ce3 = 0, a_ce2 = 0, a_kl2 = 0, a_kl3 = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    Row p3, p2, lp3, lp2;
    int am3, am2;
    load_row(l3 + i * c, c, p3);
    load_row(l2 + i * c, c, p2);
    softmax_row(p3, c, lp3, am3);
    softmax_row(p2, c, lp2, am2);
    const int64_t y = label[i];
    const bool yv = y >= 0 && y < c;
    const float w = yv ? (cw ? cw[y] : 1.f) : 0.f;
    if (yv) {
      a_ce3 += (double)(-w * lp3.v[y]);
      a_ce2 += (double)(-w * lp2.v[y]);
      if (y != ignore_index) {
        if (conf3) atomicAdd(&cf3[y * c + am3], 1u);
        if (conf2) atomicAdd(&cf2[y * c + am2], 1u);
      }
    }
    // cross-entropy gradients (weighted mean): w/W * (softmax - onehot)
    Row g;
#pragma unroll
    for (int j = 0; j < LC_MAX; ++j)
      if (j < c) g.v[j] = w * invW * (p3.v[j] - ((yv && j == y) ? 1.f : 0.f));
    Row gce3 = g;
#pragma unroll
    for (int j = 0; j < LC_MAX; ++j)
      if (j < c) g.v[j] = w * invW * (p2.v[j] - ((yv && j == y) ? 1.f : 0.f));
    Row gce2 = g;
    if (lambda_xm > 0.f) {
      // KL(softmax(other main head) || softmax(this second head)), mean over points
      Row q2, lq2, q3, lq3;
      int dummy;
      if (dual) {
        load_row(l2b + i * c, c, q2);
        load_row(l3b + i * c, c, q3);
        softmax_row(q2, c, lq2, dummy);
        softmax_row(q3, c, lq3, dummy);
      } else {
        q2 = p2; lq2 = lp2; q3 = p3; lq3 = lp3;
      }
      float kl2 = 0.f, kl3 = 0.f;
#pragma unroll
      for (int j = 0; j < LC_MAX; ++j)
        if (j < c) {
          float t3 = p3.v[j], t2 = p2.v[j];
          kl2 += t3 > 0.f ? t3 * (lp3.v[j] - lq2.v[j]) : 0.f;   // target = softmax(lidar main), input = log_softmax(img second)
          kl3 += t2 > 0.f ? t2 * (lp2.v[j] - lq3.v[j]) : 0.f;
        }
      a_kl2 += kl2;
      a_kl3 += kl3;
      const float s = lambda_xm * invN;
      if (dual) {
#pragma unroll
        for (int j = 0; j < LC_MAX; ++j)
          if (j < c) {
            g.v[j] = s * (q2.v[j] - p3.v[j]);
            q3.v[j] = s * (q3.v[j] - p2.v[j]);
          }
        store_row(g2b + i * c, c, g);
        store_row(g3b + i * c, c, q3);
      } else {
#pragma unroll
        for (int j = 0; j < LC_MAX; ++j)
          if (j < c) {
            gce2.v[j] += s * (p2.v[j] - p3.v[j]);
            gce3.v[j] += s * (p3.v[j] - p2.v[j]);
          }
      }
    } else if (dual) {
      Row z;
#pragma unroll
      for (int j = 0; j < LC_MAX; ++j) z.v[j] = 0.f;
      store_row(g2b + i * c, c, z);
      store_row(g3b + i * c, c, z);
    }
    store_row(g3 + i * c, c, gce3);
    store_row(g2 + i * c, c, gce2);
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

__global__ void loss_finalize_kernel(const double *__restrict__ part, int nb, const double *__restrict__ wsum, float ce_scale, float lambda_xm,
                                     int64_t n, float *__restrict__ losses) {
  if (threadIdx.x != 0) return;
  double s[4] = {0, 0, 0, 0};
  for (int b = 0; b < nb; ++b)
    for (int j = 0; j < 4; ++j) s[j] += part[(int64_t)b * 4 + j];
  const double W = wsum[0];
  const double ce3 = s[0] / W, ce2 = s[1] / W, kl2 = s[2] / (double)n, kl3 = s[3] / (double)n;
  losses[0] = (float)(ce_scale * ce2 + lambda_xm * kl2);   // loss_2d
  losses[1] = (float)(ce_scale * ce3 + lambda_xm * kl3);   // loss_3d
}

extern "C" size_t ftx_fusion_loss_workspace_bytes(void) { return sizeof(double) * (LOSS_BLOCKS * 4 + 2) + 256; }

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
  double *wsum = part + LOSS_BLOCKS * 4;
  loss_wsum_kernel<<<1, 1024, 0, st>>>(label, class_weights, n, c, wsum);
  const int nb = (int)(ceil_div(n, 256) < LOSS_BLOCKS ? ceil_div(n, 256) : LOSS_BLOCKS);
  loss_main_kernel<<<nb, 256, 0, st>>>(lidar_logit, img_logit, dual ? lidar_logit2 : lidar_logit, dual ? img_logit2 : img_logit, label, class_weights,
                                       wsum, ce_scale, lambda_xm, n, c, ignore_index, grad_lidar, grad_img, dual ? grad_lidar2 : grad_lidar,
                                       dual ? grad_img2 : grad_img, (long long *)conf3d, (long long *)conf2d, part);
  loss_finalize_kernel<<<1, 64, 0, st>>>(part, nb, wsum, ce_scale, lambda_xm, n, losses);
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
