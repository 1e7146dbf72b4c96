// ViT self-attention (timm Attention.forward, reached at models/transformers.py:36-37):
//   out = softmax(Q K^T * scale) V     per (batch, head), T tokens (578), head dim 64, fp32.
// Flash-style: the T x T score matrix is never written.  Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Layout trick (cdna_hip_programming.md "accumulator tile as the next MFMA's operand"): a 32x32
// accumulator has its COLUMN on the lane and 16 ROWS in registers (row(g,h) = (g&3)+8(g>>2)+4h,
// h = lane>>5).  A following MFMA that sums over the tile's ROW index can take it as the B operand
// with no lane movement.  So each product is oriented so that the index summed next is the row:
//   forward   S^T[key][q] = K Q^T      -> softmax state per lane (one q per lane)
//             O^T[dv][q] += V^T P^T    (sums over keys = rows of P^T)
//   dK/dV     S[q][key]   = Q K^T,  dP[q][key] = dO V^T
//             dV^T[dv][key] += dO^T P, dK^T[d][key] += Q^T dS      (sum over q = rows)
//   dQ        S^T, dP^T as in the forward orientation;  dQ^T[d][q] += K^T dS^T  (sum over keys)
// The reduction index of the first products is permuted (lane half h takes d = 8t+4h+s) so operand
// fragments are one ds_read_b128 / one 16-byte global load per 4 MFMAs.
#include "../../../fusiontransformer_amd/csrc/ftx_common.h"

using namespace ftx;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HD = 64;          // head dim (fixed)
constexpr int TS = 68;          // LDS row stride in floats (16-byte aligned, conflict-free b128)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ inline int acc_row(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// 2^x as the bare v_exp_f32: every argument in these kernels is <= 0 up to rounding (a score minus its row maximum / log-sum-exp)
// or -inf (a masked key), so exp2f()'s range fix-ups -- a compare, two selects and an ldexp per call, 5 of the 6 instructions --
// have nothing to fix; results below 2^-126 flush to zero, which is what a softmax weight of that size is worth.
__device__ inline float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }

// qkv (b, t, 3, nh, 64): row of tensor `which` (0 q, 1 k, 2 v) for token t, head hd
__device__ inline const float *qkv_row(const float *qkv, int b, int t, int which, int hd, int T, int nh) {
  return qkv + ((((int64_t)b * T + t) * 3 + which) * nh + hd) * HD;
}

// Load this lane's permuted 32-float fragment of a 64-float row: elements 8t+4h+s.
__device__ inline void load_frag(const float *row, int h, bool valid, float (&f)[32]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *(const float4 *)(row + 8 * t + 4 * h);
    f[4 * t + 0] = v.x; f[4 * t + 1] = v.y; f[4 * t + 2] = v.z; f[4 * t + 3] = v.w;
  }
}

// acc[row][col=lane] += sum_d Lds[row][d] * frag[d]   (A operand from LDS rows, B operand = lane's fragment)
__device__ inline void mfma_lds_x_frag(const float *lds_tile, int l31, int h, const float (&frag)[32], f32x16 &acc) {
  const float *rp = lds_tile + l31 * TS + 4 * h;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float4 a = *(const float4 *)(rp + 8 * t);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, frag[4 * t + 0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, frag[4 * t + 1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, frag[4 * t + 2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, frag[4 * t + 3], acc, 0, 0, 0);
  }
}

// acc[row][col=lane] += sum_d frag[d] * Lds[col][d]   (A operand = lane's fragment as a row, B from LDS rows)
__device__ inline void mfma_frag_x_lds(const float (&frag)[32], const float *lds_tile, int l31, int h, f32x16 &acc) {
  const float *rp = lds_tile + l31 * TS + 4 * h;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float4 b = *(const float4 *)(rp + 8 * t);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[4 * t + 0], b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[4 * t + 1], b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[4 * t + 2], b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[4 * t + 3], b.w, acc, 0, 0, 0);
  }
}

// out[r][col=lane] += sum over the 32 rows j of X of  Lds[j][r_off + r] * X[j][col]
// (X = an accumulator tile used as the B operand; A operand = column slice of an LDS tile)
__device__ inline void mfma_ldsT_x_acc(const float *lds_tile, int col_off, int l31, int h, const f32x16 &x, f32x16 &out) {
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float a = lds_tile[acc_row(g, h) * TS + col_off + l31];
    out = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[g], out, 0, 0, 0);
  }
}

// Stage a 32-row x 64-float tile (rows t0.. of tensor `which`) into LDS with stride TS; rows >= T are zero.  NT = threads of the
// staging group (one key / query group of a block), tid in [0, NT).
template <int NT>
__device__ inline void tile_prefetch(const float *qkv, int b, int hd, int which, int t0, int T, int nh, int tid, float4 (&r)[512 / NT]) {
#pragma unroll
  for (int q = 0; q < 512 / NT; ++q) {
    int e = q * NT + tid;
    int row = e >> 4, c4 = (e & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + row < T) v = *(const float4 *)(qkv_row(qkv, b, t0 + row, which, hd, T, nh) + c4);
    r[q] = v;
  }
}
template <int NT>
__device__ inline void tile_store(float *lds_tile, int tid, const float4 (&r)[512 / NT]) {
#pragma unroll
  for (int q = 0; q < 512 / NT; ++q) {
    int e = q * NT + tid;
    int row = e >> 4, c4 = (e & 15) * 4;
    *(float4 *)&lds_tile[row * TS + c4] = r[q];
  }
}
// Same for a (b, t, nh*64) tensor (out / grad_out): head slice of 64 floats per token.
template <int NT>
__device__ inline void tile_prefetch_o(const float *o, int b, int hd, int t0, int T, int nh, int tid, float4 (&r)[512 / NT]) {
#pragma unroll
  for (int q = 0; q < 512 / NT; ++q) {
    int e = q * NT + tid;
    int row = e >> 4, c4 = (e & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + row < T) v = *(const float4 *)(o + (((int64_t)b * T + t0 + row) * nh + hd) * HD + c4);
    r[q] = v;
  }
}

// ---------------------------------------------------------------------------------------
// All three kernels: block = QW waves of 32 queries (or keys) x SPLIT groups.  Group g walks the inner tiles g, g+SPLIT, ... with
// its own LDS tile pair; the groups' partial results are merged pairwise through LDS at the end (a fixed tree: 0<-1, 2<-3, ..., 0<-2,
// ..., so the result depends on (QW, SPLIT) only, never on timing).  578 tokens give 19 wave-tiles per (frame, head): 228 per layer at
// batch 1, 912 at batch 4, for 1024 SIMDs -- so the launcher picks (QW, SPLIT) from the number of wave-tiles: few of them => one wave
// per group and up to 8 key groups (the serial key loop, which is what bounds a small launch, gets 4x shorter), many => 4 waves
// sharing each staged tile.  With QW = 1 a group is one wave, which orders its own LDS traffic: no block barrier in the loop.
// ---------------------------------------------------------------------------------------
template <int QW>
__device__ inline void group_sync() {
  if (QW == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }
// floats of LDS: SPLIT tile pairs (+ per-tile row statistics), reused after the loop for SPLIT/2 merge slots of QW x NREG x 64 floats
template <int QW, int SPLIT, int NREG, int EXTRA>
constexpr int smem_floats() { return cmax(SPLIT * (2 * 32 * TS + EXTRA), cmax(SPLIT / 2, 1) * QW * NREG * 64); }

// Sum the groups' accumulator tiles into group 0: tree over the groups, fixed order.
template <int QW, int SPLIT, int NTILES>
__device__ inline void merge_sum(float *smem, int grp, int wave, int lane, f32x16 (&acc)[NTILES]) {
  if (SPLIT == 1) return;
#pragma unroll
  for (int s = 1; s < SPLIT; s <<= 1) {
    float *cw = smem + ((grp / (2 * s)) * QW + wave) * (16 * NTILES) * 64 + lane;
    __syncthreads();   // tiles (round 1) / the slot's previous contents are dead
    if ((grp & (2 * s - 1)) == s) {
#pragma unroll
      for (int t = 0; t < NTILES; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) cw[(t * 16 + g) * 64] = acc[t][g];
    }
    __syncthreads();
    if ((grp & (2 * s - 1)) == 0 && grp + s < SPLIT) {
#pragma unroll
      for (int t = 0; t < NTILES; ++t)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[t][g] += cw[(t * 16 + g) * 64];
    }
  }
}

// forward: wave = 32 queries; loop over 32-key tiles
template <int QW, int SPLIT>
__global__ __launch_bounds__(64 * QW * SPLIT) void attn_fwd_kernel(const float *__restrict__ qkv, int T, int nh, float scale,
                                                                   float *__restrict__ out, float *__restrict__ lse, unsigned long long *__restrict__ stamps) {
  const bool rec = blockIdx.x == 1 && blockIdx.y == 3 && blockIdx.z == 1 && (threadIdx.x & 63) == 0;
  unsigned long long *sp = stamps + (threadIdx.x >> 6) * 128;
  int si = 0;
#define STAMP() do { if (rec && si < 128) sp[si++] = __builtin_amdgcn_s_memtime(); } while (0)
  STAMP();
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) float smem[smem_floats<QW, SPLIT, 34, 0>()];
  const int tid = threadIdx.x % NT, grp = threadIdx.x / NT;
  float *Ks = smem + grp * (2 * 32 * TS), *Vs = Ks + 32 * TS;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * (32 * QW) + wave * 32 + l31;   // this lane's query (the accumulator column)
  const bool qv = q < T;
  const float sl2 = scale * LOG2E;

  float qf[32];
  load_frag(qkv_row(qkv, b, qv ? q : 0, 0, hd, T, nh), h, qv, qf);
#pragma unroll
  for (int i = 0; i < 32; ++i) qf[i] *= sl2;   // scale * log2(e) folded into Q once: S^T comes out of the MFMAs in the exp2 domain

  f32x16 o0, o1;
#pragma unroll
  for (int g = 0; g < 16; ++g) o0[g] = o1[g] = 0.f;
  float m = -INFINITY, l = 0.f;

  const int ntiles = (T + 31) / 32;
  const int iters = (ntiles + SPLIT - 1) / SPLIT;
  float4 rk[512 / NT], rv[512 / NT];
  tile_prefetch<NT>(qkv, b, hd, 1, grp * 32, T, nh, tid, rk);
  tile_prefetch<NT>(qkv, b, hd, 2, grp * 32, T, nh, tid, rv);
  for (int it = 0; it < iters; ++it) {
    const int kt = it * SPLIT + grp;
    STAMP();
    tile_store<NT>(Ks, tid, rk);
    tile_store<NT>(Vs, tid, rv);
    STAMP();
    group_sync<QW>();
    STAMP();
    if (it + 1 < iters) {
      tile_prefetch<NT>(qkv, b, hd, 1, (kt + SPLIT) * 32, T, nh, tid, rk);
      tile_prefetch<NT>(qkv, b, hd, 2, (kt + SPLIT) * 32, T, nh, tid, rv);
    }
    if (kt < ntiles) {
      // S^T[key][q] * scale * log2(e): rows = keys of this tile, column = this lane's query
      f32x16 st;
#pragma unroll
      for (int g = 0; g < 16; ++g) st[g] = 0.f;
      mfma_lds_x_frag(Ks, l31, h, qf, st);
      asm volatile("" :: "v"(st[0]), "v"(st[15]));
      STAMP();
      if (kt == ntiles - 1) {   // only the last tile has keys past T (wave-uniform branch)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          if (kt * 32 + acc_row(g, h) >= T) st[g] = -INFINITY;
      }
      float mx = fmaxf(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])), fmaxf(fmaxf(st[4], st[5]), fmaxf(st[6], st[7])));
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(st[8], st[9]), fmaxf(st[10], st[11])), fmaxf(fmaxf(st[12], st[13]), fmaxf(st[14], st[15]))));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));     // the other 16 keys of the tile live in the partner half
      const float m_new = fmaxf(m, mx);
      float rs = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float p = exp2_raw(st[g] - m_new);
        st[g] = p;
        rs += p;
      }
      rs += __shfl_xor(rs, 32, 64);
      if (__any(m_new != m)) {   // the running maximum moved for some query of this wave: rescale (a factor of exactly 1 is skipped)
        const float alpha = exp2_raw(m - m_new);
        l *= alpha;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          o0[g] *= alpha;
          o1[g] *= alpha;
        }
      }
      l += rs;
      m = m_new;
      STAMP();
      // O^T[dv][q] += sum_key V[key][dv] * P^T[key][q]
      mfma_ldsT_x_acc(Vs, 0, l31, h, st, o0);
      mfma_ldsT_x_acc(Vs, 32, l31, h, st, o1);
      asm volatile("" :: "v"(o0[0]), "v"(o1[15]));
      STAMP();
    }
    group_sync<QW>();
    STAMP();
  }
  if (SPLIT > 1) {   // merge the groups' (m, l, O): tree in fixed order
#pragma unroll
    for (int s = 1; s < SPLIT; s <<= 1) {
      float *cw = smem + ((grp / (2 * s)) * QW + wave) * 34 * 64 + lane;
      __syncthreads();
      if ((grp & (2 * s - 1)) == s) {
        cw[0] = m;
        cw[64] = l;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          cw[(2 + g) * 64] = o0[g];
          cw[(18 + g) * 64] = o1[g];
        }
      }
      __syncthreads();
      if ((grp & (2 * s - 1)) == 0 && grp + s < SPLIT) {
        const float m1 = cw[0], l1 = cw[64];
        const float mt = fmaxf(m, m1);
        const float a0 = (m == -INFINITY) ? 0.f : exp2f(m - mt), a1 = (m1 == -INFINITY) ? 0.f : exp2f(m1 - mt);
        l = l * a0 + l1 * a1;
        m = mt;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          o0[g] = o0[g] * a0 + cw[(2 + g) * 64] * a1;
          o1[g] = o1[g] * a0 + cw[(18 + g) * 64] * a1;
        }
      }
    }
  }
  if (qv && grp == 0) {
    const float inv = 1.f / l;
    float *op = out + (((int64_t)b * T + q) * nh + hd) * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      int r = acc_row(4 * g4, h);
      *(float4 *)(op + r) = make_float4(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
      *(float4 *)(op + 32 + r) = make_float4(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
    }
    if (h == 0) lse[((int64_t)b * nh + hd) * T + q] = (m + log2f(l)) * LN2;   // ln sum_k exp(scale * s)
  }
}


#include <vector>
#include <cstdio>
int main() {
  const int B = 4, T = 578, H = 12;
  size_t nq = (size_t)B * T * 3 * H * 64;
  float *qkv, *out, *lse; unsigned long long *st;
  hipMalloc(&qkv, nq * 4); hipMalloc(&out, (size_t)B * T * H * 64 * 4); hipMalloc(&lse, (size_t)B * H * T * 4); hipMalloc(&st, 8 * 128 * 8);
  std::vector<float> h(nq); for (size_t i = 0; i < nq; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice); hipMemset(st, 0, 8 * 128 * 8);
  dim3 grid((T + 127) / 128, H, B);
  for (int r = 0; r < 3; ++r) attn_fwd_kernel<4, 2><<<grid, 512>>>(qkv, T, H, 0.125f, out, lse, st);
  hipDeviceSynchronize();
  std::vector<unsigned long long> s(8 * 128); hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
  for (int w : {0, 4}) {
    printf("wave %d (group %d): start->first store %llu\n", w, w / 4, s[w * 128 + 1] - s[w * 128]);
    printf(" it | store  sync1   S-mfma softmax O-mfma  sync2 | total   (s_memtime ticks, 100 MHz: x21 for core cycles at 2.1 GHz)\n");
    for (int it = 0; it < 10; ++it) {
      unsigned long long *p = &s[w * 128 + 1 + it * 7];
      if (p[6] == 0) break;
      printf(" %2d | %5llu %6llu %7llu %6llu %6llu %6llu | %6llu\n", it, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4], p[6] - p[5], p[6] - p[0]);
    }
  }
  return 0;
}
