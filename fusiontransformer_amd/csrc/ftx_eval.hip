// Evaluation scatter-back: predictions of the model points (voxels) mapped to the ORIGINAL points of each
// frame through `inverse_map`, label ids mapped back, confusion matrices updated -- one pass, no host copies.
// Reference: FusionTransformer/data/utils/validate.py:62-120 (argmax / softmax-sum ensemble,
// map_sparse_to_org, map_inverse_label) and data/utils/evaluate.py:12-26 (Evaluator.update).
#include "ftx_common.h"
using namespace ftx;

constexpr int EV_MAX_C = 32;

__global__ __launch_bounds__(256) void eval_scatter_back_kernel(const float *__restrict__ l3, const float *__restrict__ l2, int64_t n_rows, int c,
                                                                const int64_t *__restrict__ inverse, const int32_t *__restrict__ gt, int64_t m,
                                                                const int32_t *__restrict__ class_labels, int32_t *__restrict__ pred3,
                                                                int32_t *__restrict__ pred2, int32_t *__restrict__ prede,
                                                                unsigned long long *__restrict__ conf3, unsigned long long *__restrict__ conf2,
                                                                unsigned long long *__restrict__ confe, int32_t *__restrict__ bad) {
  __shared__ int h3[EV_MAX_C * EV_MAX_C], h2[EV_MAX_C * EV_MAX_C], he[EV_MAX_C * EV_MAX_C];
  __shared__ int s_lab[EV_MAX_C];
  __shared__ int s_ignore_idx;  // index of label id `c` (what Evaluator.update turns gt == 0 into) in class_labels, or -1
  const int tid = threadIdx.x;
  for (int i = tid; i < c * c; i += 256) { h3[i] = 0; h2[i] = 0; he[i] = 0; }
  if (tid < c) s_lab[tid] = class_labels[tid];
  if (tid == 0) {
    int idx = -1;
    for (int j = 0; j < c; ++j)
      if (class_labels[j] == c && idx < 0) idx = j;
    s_ignore_idx = idx;
  }
  __syncthreads();
  for (int64_t i = blockIdx.x * (int64_t)256 + tid; i < m; i += (int64_t)gridDim.x * 256) {
    const int64_t r = inverse[i];
    const int g = gt[i];
    if (r < 0 || r >= n_rows || g < 0 || g >= c) {
      atomicExch(bad, 1);
      continue;
    }
    int p3 = -1, p2 = -1, pe = -1;
    float m3 = -INFINITY, m2 = -INFINITY;
    if (l3) {
      const float *row = l3 + r * c;
      for (int j = 0; j < c; ++j) { float v = row[j]; if (v > m3) { m3 = v; p3 = j; } }
    }
    if (l2) {
      const float *row = l2 + r * c;
      for (int j = 0; j < c; ++j) { float v = row[j]; if (v > m2) { m2 = v; p2 = j; } }
    }
    if (l3 && l2) {  // (softmax(2d) + softmax(3d)).argmax(1), first maximum wins
      const float *r3 = l3 + r * c, *r2 = l2 + r * c;
      float s3 = 0.f, s2 = 0.f;
      for (int j = 0; j < c; ++j) { s3 += expf(r3[j] - m3); s2 += expf(r2[j] - m2); }
      float best = -INFINITY;
      for (int j = 0; j < c; ++j) {
        float v = expf(r2[j] - m2) / s2 + expf(r3[j] - m3) / s3;
        if (v > best) { best = v; pe = j; }
      }
    }
    // gt: learning id -> original id; id 0 becomes `num_classes` (evaluate.py:22), which only counts if it is a label id
    int go = s_lab[g];
    int gi = g;
    if (go == 0) gi = s_ignore_idx;
    if (pred3 && p3 >= 0) pred3[i] = s_lab[p3];
    if (pred2 && p2 >= 0) pred2[i] = s_lab[p2];
    if (prede && pe >= 0) prede[i] = s_lab[pe];
    if (gi >= 0) {
      if (p3 >= 0) atomicAdd(&h3[gi * c + p3], 1);
      if (p2 >= 0) atomicAdd(&h2[gi * c + p2], 1);
      if (pe >= 0) atomicAdd(&he[gi * c + pe], 1);
    }
  }
  __syncthreads();
  for (int i = tid; i < c * c; i += 256) {
    if (conf3 && h3[i]) atomicAdd(&conf3[i], (unsigned long long)h3[i]);
    if (conf2 && h2[i]) atomicAdd(&conf2[i], (unsigned long long)h2[i]);
    if (confe && he[i]) atomicAdd(&confe[i], (unsigned long long)he[i]);
  }
}

extern "C" int ftx_eval_scatter_back(const float *logits3d, const float *logits2d, int64_t n_rows, int32_t num_classes, const int64_t *inverse,
                                     const int32_t *gt, int64_t m, const int32_t *class_labels, int32_t *pred3d, int32_t *pred2d,
                                     int32_t *pred_ens, int64_t *conf3d, int64_t *conf2d, int64_t *conf_ens, int32_t *bad_flag, void *stream) {
  FTX_REQUIRE(n_rows >= 0 && m >= 0, "ftx_eval_scatter_back: negative size");
  FTX_REQUIRE(num_classes >= 1 && num_classes <= EV_MAX_C, "ftx_eval_scatter_back: num_classes %d outside 1..%d", num_classes, EV_MAX_C);
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(logits3d || logits2d, "ftx_eval_scatter_back: no logits");
  FTX_REQUIRE(inverse && gt && class_labels && bad_flag, "ftx_eval_scatter_back: null pointer");
  hipStream_t st = (hipStream_t)stream;
  int64_t g = ceil_div(m, 256 * 8);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  eval_scatter_back_kernel<<<(unsigned)g, 256, 0, st>>>(logits3d, logits2d, n_rows, num_classes, inverse, gt, m, class_labels, pred3d, pred2d, pred_ens,
                                                       (unsigned long long *)conf3d, (unsigned long long *)conf2d, (unsigned long long *)conf_ens,
                                                       bad_flag);
  return check_launch("ftx_eval_scatter_back");
}

// ---------------------------------------------------------------------------------------
// Offline LiDAR -> image projection (data/semantic_kitti/preprocess.py:108-116): homogeneous point times the 3x4
// float32 matrix P2 * Tr, perspective divide, frustum test 0 < u < width, 0 < v < height.  keep[i] = 1 for
// points in front of the vehicle (x > 0) that land inside the image; rowcol[i] = (v, u) (the reference's fliplr).
// The dot product is accumulated k = 0..3 with fused multiply-adds, the order of the host BLAS the reference
// calls through numpy (the golden vectors of tests/golden/projection.npz are reproduced bit for bit).
// ---------------------------------------------------------------------------------------
__global__ void project_points_kernel(const float *__restrict__ pts, int64_t n, const float *__restrict__ P, float width, float height,
                                      uint8_t *__restrict__ keep, float *__restrict__ rowcol) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    float h[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float acc = P[4 * r] * x;
      acc = fmaf(P[4 * r + 1], y, acc);
      acc = fmaf(P[4 * r + 2], z, acc);
      acc = fmaf(P[4 * r + 3], 1.0f, acc);
      h[r] = acc;
    }
    const float u = h[0] / h[2], v = h[1] / h[2];
    const bool in = (x > 0.f) && (u > 0.f) && (v > 0.f) && (u < width) && (v < height);
    keep[i] = in ? 1 : 0;
    rowcol[2 * i] = v;
    rowcol[2 * i + 1] = u;
  }
}

extern "C" int ftx_project_points(const float *points, int64_t n, const float *proj_matrix, int32_t width, int32_t height, uint8_t *keep,
                                  float *rowcol, void *stream) {
  FTX_REQUIRE(n >= 0 && width > 0 && height > 0, "ftx_project_points: bad size");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(points && proj_matrix && keep && rowcol, "ftx_project_points: null pointer");
  project_points_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(points, n, proj_matrix, (float)width, (float)height, keep, rowcol);
  return check_launch("ftx_project_points");
}
