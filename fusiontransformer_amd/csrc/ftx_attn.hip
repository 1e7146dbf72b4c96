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
#include "ftx_common.h"

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
                                                                   float *__restrict__ out, float *__restrict__ lse) {
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) float smem[smem_floats<QW, SPLIT, 34, 0>()];
  const int tid = threadIdx.x % NT, grp = threadIdx.x / NT;
  float *Ks = smem + grp * (2 * 32 * TS), *Vs = Ks + 32 * TS;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * (32 * QW) + wave * 32 + l31;   // this lane's query (the accumulator column)
  const bool qv = q < T;
  // 578 tokens = 18 wave-tiles + 2 rows: with 4 waves per block the 20th wave-tile holds no query at all.  Such a wave still stages
  // tiles and keeps the barriers, but skips its MFMAs and softmax (wave-uniform), leaving the SIMD to the waves beside it.
  const bool wave_live = blockIdx.x * (32 * QW) + wave * 32 < T;
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
    tile_store<NT>(Ks, tid, rk);
    tile_store<NT>(Vs, tid, rv);
    group_sync<QW>();
    if (it + 1 < iters) {
      tile_prefetch<NT>(qkv, b, hd, 1, (kt + SPLIT) * 32, T, nh, tid, rk);
      tile_prefetch<NT>(qkv, b, hd, 2, (kt + SPLIT) * 32, T, nh, tid, rv);
    }
    if (kt < ntiles && wave_live) {
      // S^T[key][q] * scale * log2(e): rows = keys of this tile, column = this lane's query
      f32x16 st;
#pragma unroll
      for (int g = 0; g < 16; ++g) st[g] = 0.f;
      mfma_lds_x_frag(Ks, l31, h, qf, st);
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
      // O^T[dv][q] += sum_key V[key][dv] * P^T[key][q]
      mfma_ldsT_x_acc(Vs, 0, l31, h, st, o0);
      mfma_ldsT_x_acc(Vs, 32, l31, h, st, o1);
    }
    group_sync<QW>();
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

// ---------------------------------------------------------------------------------------
// backward helpers
// ---------------------------------------------------------------------------------------
// delta[b,h,t] = sum_dv dO[b,t,h,dv] * O[b,t,h,dv]
__global__ void attn_delta_kernel(const float *__restrict__ o, const float *__restrict__ go, int64_t rows, int T, int nh, float *__restrict__ delta) {
  // one 16-lane group per (b, t, h) row of 64 floats
  const int64_t gid = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  if (gid >= rows) return;
  float4 a = *(const float4 *)(o + gid * HD + sub * 4);
  float4 g = *(const float4 *)(go + gid * HD + sub * 4);
  float s = a.x * g.x + a.y * g.y + a.z * g.z + a.w * g.w;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
  if (sub == 0) {
    int64_t bt = gid / nh;
    int hd = (int)(gid - bt * nh);
    int64_t bb = bt / T;
    int t = (int)(bt - bb * T);
    delta[(bb * nh + hd) * T + t] = s;
  }
}

// dK, dV: wave = 32 keys; loop over 32-query tiles
template <int QW, int SPLIT>
__global__ __launch_bounds__(64 * QW * SPLIT) void attn_bwd_kv_kernel(const float *__restrict__ qkv, const float *__restrict__ go,
                                                                      const float *__restrict__ lse, const float *__restrict__ delta, int T,
                                                                      int nh, float scale, float *__restrict__ gqkv) {
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) float smem[smem_floats<QW, SPLIT, 64, 64>()];
  const int tid = threadIdx.x % NT, grp = threadIdx.x / NT;
  float *Qs = smem + grp * (2 * 32 * TS + 64), *Gs = Qs + 32 * TS, *s_lse = Gs + 32 * TS, *s_delta = s_lse + 32;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int key = blockIdx.x * (32 * QW) + wave * 32 + l31;   // accumulator column = this lane's key
  const bool kv = key < T;
  const bool wave_live = blockIdx.x * (32 * QW) + wave * 32 < T;   // a wave-tile with no key at all skips its products (see attn_fwd_kernel)
  const float sl2 = scale * LOG2E;

  float kf[32], vf[32];
  load_frag(qkv_row(qkv, b, kv ? key : 0, 1, hd, T, nh), h, kv, kf);
  load_frag(qkv_row(qkv, b, kv ? key : 0, 2, hd, T, nh), h, kv, vf);
#pragma unroll
  for (int i = 0; i < 32; ++i) kf[i] *= sl2;   // only S uses this fragment: S comes out of the MFMAs in the exp2 domain

  f32x16 acc[4];   // dV^T tile 0/1, dK^T tile 0/1
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[t][g] = 0.f;

  const int ntiles = (T + 31) / 32;
  const int iters = (ntiles + SPLIT - 1) / SPLIT;
  float4 rq[512 / NT], rg[512 / NT];
  tile_prefetch<NT>(qkv, b, hd, 0, grp * 32, T, nh, tid, rq);
  tile_prefetch_o<NT>(go, b, hd, grp * 32, T, nh, tid, rg);
  for (int it = 0; it < iters; ++it) {
    const int qt = it * SPLIT + grp;
    tile_store<NT>(Qs, tid, rq);
    tile_store<NT>(Gs, tid, rg);
    if (tid < 32) {
      int t = qt * 32 + tid;
      s_lse[tid] = t < T ? lse[((int64_t)b * nh + hd) * T + t] * LOG2E : 0.f;
      s_delta[tid] = t < T ? delta[((int64_t)b * nh + hd) * T + t] : 0.f;
    }
    group_sync<QW>();
    if (it + 1 < iters) {
      tile_prefetch<NT>(qkv, b, hd, 0, (qt + SPLIT) * 32, T, nh, tid, rq);
      tile_prefetch_o<NT>(go, b, hd, (qt + SPLIT) * 32, T, nh, tid, rg);
    }
    if (qt < ntiles && wave_live) {
      // S[q][key] and dP[q][key]: rows = queries of the tile, column = this lane's key
      f32x16 s, dp;
#pragma unroll
      for (int g = 0; g < 16; ++g) s[g] = dp[g] = 0.f;
      mfma_lds_x_frag(Qs, l31, h, kf, s);
      mfma_lds_x_frag(Gs, l31, h, vf, dp);
      // P = exp2(S - lse), dS = P (dP - delta) scale.  A key lane past T holds zero K / V fragments: its column is finite garbage
      // that is never stored; query rows past T exist in the last tile only (wave-uniform branch).
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int r = acc_row(g, h);
        const float p = exp2_raw(s[g] - s_lse[r]);
        s[g] = p;                                            // P
        dp[g] = p * (dp[g] - s_delta[r]) * scale;             // dS
      }
      if (qt == ntiles - 1) {
#pragma unroll
        for (int g = 0; g < 16; ++g)
          if (qt * 32 + acc_row(g, h) >= T) s[g] = dp[g] = 0.f;
      }
      // dV^T[dv][key] += sum_q dO[q][dv] P[q][key];  dK^T[d][key] += sum_q Q[q][d] dS[q][key]
      mfma_ldsT_x_acc(Gs, 0, l31, h, s, acc[0]);
      mfma_ldsT_x_acc(Gs, 32, l31, h, s, acc[1]);
      mfma_ldsT_x_acc(Qs, 0, l31, h, dp, acc[2]);
      mfma_ldsT_x_acc(Qs, 32, l31, h, dp, acc[3]);
    }
    group_sync<QW>();
  }
  merge_sum<QW, SPLIT, 4>(smem, grp, wave, lane, acc);
  if (kv && grp == 0) {
    float *kp = gqkv + ((((int64_t)b * T + key) * 3 + 1) * nh + hd) * HD;
    float *vp = gqkv + ((((int64_t)b * T + key) * 3 + 2) * nh + hd) * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      int r = acc_row(4 * g4, h);
      *(float4 *)(vp + r) = make_float4(acc[0][4 * g4], acc[0][4 * g4 + 1], acc[0][4 * g4 + 2], acc[0][4 * g4 + 3]);
      *(float4 *)(vp + 32 + r) = make_float4(acc[1][4 * g4], acc[1][4 * g4 + 1], acc[1][4 * g4 + 2], acc[1][4 * g4 + 3]);
      *(float4 *)(kp + r) = make_float4(acc[2][4 * g4], acc[2][4 * g4 + 1], acc[2][4 * g4 + 2], acc[2][4 * g4 + 3]);
      *(float4 *)(kp + 32 + r) = make_float4(acc[3][4 * g4], acc[3][4 * g4 + 1], acc[3][4 * g4 + 2], acc[3][4 * g4 + 3]);
    }
  }
}

// dQ: wave = 32 queries; loop over 32-key tiles
template <int QW, int SPLIT>
__global__ __launch_bounds__(64 * QW * SPLIT) void attn_bwd_q_kernel(const float *__restrict__ qkv, const float *__restrict__ go,
                                                                     const float *__restrict__ lse, const float *__restrict__ delta, int T,
                                                                     int nh, float scale, float *__restrict__ gqkv) {
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) float smem[smem_floats<QW, SPLIT, 32, 0>()];
  const int tid = threadIdx.x % NT, grp = threadIdx.x / NT;
  float *Ks = smem + grp * (2 * 32 * TS), *Vs = Ks + 32 * TS;
  const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * (32 * QW) + wave * 32 + l31;
  const bool qv = q < T;
  const bool wave_live = blockIdx.x * (32 * QW) + wave * 32 < T;   // a wave-tile with no query at all skips its products (see attn_fwd_kernel)
  const float sl2 = scale * LOG2E;

  float qf[32], gf[32];
  load_frag(qkv_row(qkv, b, qv ? q : 0, 0, hd, T, nh), h, qv, qf);
#pragma unroll
  for (int i = 0; i < 32; ++i) qf[i] *= sl2;   // only S^T uses this fragment
  load_frag(go + (((int64_t)b * T + (qv ? q : 0)) * nh + hd) * HD, h, qv, gf);
  const float my_lse = qv ? lse[((int64_t)b * nh + hd) * T + q] * LOG2E : 0.f;
  const float my_delta = qv ? delta[((int64_t)b * nh + hd) * T + q] : 0.f;

  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[0][g] = acc[1][g] = 0.f;

  const int ntiles = (T + 31) / 32;
  const int iters = (ntiles + SPLIT - 1) / SPLIT;
  float4 rk[512 / NT], rv[512 / NT];
  tile_prefetch<NT>(qkv, b, hd, 1, grp * 32, T, nh, tid, rk);
  tile_prefetch<NT>(qkv, b, hd, 2, grp * 32, T, nh, tid, rv);
  for (int it = 0; it < iters; ++it) {
    const int kt = it * SPLIT + grp;
    tile_store<NT>(Ks, tid, rk);
    tile_store<NT>(Vs, tid, rv);
    group_sync<QW>();
    if (it + 1 < iters) {
      tile_prefetch<NT>(qkv, b, hd, 1, (kt + SPLIT) * 32, T, nh, tid, rk);
      tile_prefetch<NT>(qkv, b, hd, 2, (kt + SPLIT) * 32, T, nh, tid, rv);
    }
    if (kt < ntiles && wave_live) {
      // S^T[key][q], dP^T[key][q]
      f32x16 st, dpt;
#pragma unroll
      for (int g = 0; g < 16; ++g) st[g] = dpt[g] = 0.f;
      mfma_lds_x_frag(Ks, l31, h, qf, st);
      mfma_lds_x_frag(Vs, l31, h, gf, dpt);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float p = exp2_raw(st[g] - my_lse);
        dpt[g] = p * (dpt[g] - my_delta) * scale;   // dS^T
      }
      if (kt == ntiles - 1) {   // key rows past T: only in the last tile (their K rows are zero, so P would be exp2(-lse), not 0)
#pragma unroll
        for (int g = 0; g < 16; ++g)
          if (kt * 32 + acc_row(g, h) >= T) dpt[g] = 0.f;
      }
      // dQ^T[d][q] += sum_key K[key][d] dS^T[key][q]
      mfma_ldsT_x_acc(Ks, 0, l31, h, dpt, acc[0]);
      mfma_ldsT_x_acc(Ks, 32, l31, h, dpt, acc[1]);
    }
    group_sync<QW>();
  }
  merge_sum<QW, SPLIT, 2>(smem, grp, wave, lane, acc);
  if (qv && grp == 0) {
    float *qp = gqkv + ((((int64_t)b * T + q) * 3 + 0) * nh + hd) * HD;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      int r = acc_row(4 * g4, h);
      *(float4 *)(qp + r) = make_float4(acc[0][4 * g4], acc[0][4 * g4 + 1], acc[0][4 * g4 + 2], acc[0][4 * g4 + 3]);
      *(float4 *)(qp + 32 + r) = make_float4(acc[1][4 * g4], acc[1][4 * g4 + 1], acc[1][4 * g4 + 2], acc[1][4 * g4 + 3]);
    }
  }
}

static int attn_check(const char *who, int b, int t, int h, int d) {
  FTX_REQUIRE(b >= 1 && t >= 1 && h >= 1, "%s: bad size", who);
  FTX_REQUIRE(d == HD, "%s: head dim must be 64 (got %d)", who, d);
  FTX_REQUIRE(b <= 65535 && h <= 65535, "%s: batch / heads exceed the grid limits", who);
  return FTX_OK;
}

// (QW, SPLIT) of a launch: an explicit, built tiling from the caller (ftx_attn_fwd_tiled / ftx_attn_bwd_tiled: tests and tools/bench_attn.py
// walk them), or (0, 0) = chosen here from the number of 32-row wave-tiles against the SIMDs of an MI355X (256 CUs x 4; a constant, not a
// device query: the choice -- hence nothing a caller can observe but time -- must not depend on the environment).  No process-wide state.
// Measured at 578 tokens x 12 heads (tools/bench_attn.py, us fwd / bwd): batch 1 (228 wave-tiles) (4,2) 52 / 170, (1,8) 21 / 75,
// (1,4) 23 / 70; batch 2 (4,2) 52 / 170, (2,4) 34 / 103; batch 4 (4,2) 54 / 174, (2,4) 65 / 199; batch 8 (4,2) 106 / 342, others slower.
static bool attn_tiling_built(int qw, int split) {
  return (qw == 0 && split == 0) || (qw == 4 && split == 2) || (qw == 2 && (split == 2 || split == 4)) ||
         (qw == 1 && (split == 2 || split == 4 || split == 8));
}
static void attn_config(int b, int t, int h, bool backward, int &qw, int &split) {
  if (qw != 0) return;   // explicit
  const int64_t wave_tiles = (int64_t)ceil_div(t, 32) * h * b;
  const int64_t simds = 4 * 256;
  if (wave_tiles * 4 <= simds) { qw = 1; split = backward ? 4 : 8; }   // a quarter of the SIMDs or fewer: shortest key loop
  else if (wave_tiles * 2 <= simds) { qw = 2; split = 4; }
  else { qw = 4; split = 2; }
}

#define ATTN_DISPATCH(KERNEL, ...)                                                                               \
  do {                                                                                                           \
    dim3 grid((unsigned)ceil_div(t, 32 * qw), (unsigned)h, (unsigned)b);                                         \
    if (qw == 4) KERNEL<4, 2><<<grid, 512, 0, st>>>(__VA_ARGS__);                                                \
    else if (qw == 2 && split == 2) KERNEL<2, 2><<<grid, 256, 0, st>>>(__VA_ARGS__);                             \
    else if (qw == 2) KERNEL<2, 4><<<grid, 512, 0, st>>>(__VA_ARGS__);                                           \
    else if (split == 2) KERNEL<1, 2><<<grid, 128, 0, st>>>(__VA_ARGS__);                                        \
    else if (split == 4) KERNEL<1, 4><<<grid, 256, 0, st>>>(__VA_ARGS__);                                        \
    else KERNEL<1, 8><<<grid, 512, 0, st>>>(__VA_ARGS__);                                                        \
  } while (0)

extern "C" int ftx_attn_fwd_tiled(const float *qkv, int32_t b, int32_t t, int32_t h, int32_t d, float scale, float *out, float *lse, int32_t qw,
                                  int32_t split, void *stream) {
  int rc = attn_check("ftx_attn_fwd", b, t, h, d);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(qkv && out && lse, "ftx_attn_fwd: null pointer");
  FTX_REQUIRE(attn_tiling_built(qw, split), "ftx_attn_fwd_tiled: (%d, %d) is not a built tiling", qw, split);
  hipStream_t st = (hipStream_t)stream;
  attn_config(b, t, h, false, qw, split);
  ATTN_DISPATCH(attn_fwd_kernel, qkv, t, h, scale, out, lse);
  return check_launch("ftx_attn_fwd");
}

extern "C" int ftx_attn_fwd(const float *qkv, int32_t b, int32_t t, int32_t h, int32_t d, float scale, float *out, float *lse, void *stream) {
  return ftx_attn_fwd_tiled(qkv, b, t, h, d, scale, out, lse, 0, 0, stream);
}

extern "C" size_t ftx_attn_bwd_workspace_bytes(int32_t b, int32_t t, int32_t h) {
  if (b <= 0 || t <= 0 || h <= 0) return 256;
  return sizeof(float) * (size_t)b * t * h + 256;
}

extern "C" int ftx_attn_bwd_tiled(const float *qkv, const float *out, const float *grad_out, const float *lse, int32_t b, int32_t t, int32_t h,
                                  int32_t d, float scale, float *grad_qkv, void *workspace, size_t workspace_bytes, int32_t qw, int32_t split,
                                  void *stream) {
  int rc = attn_check("ftx_attn_bwd", b, t, h, d);
  if (rc != FTX_OK) return rc;
  FTX_REQUIRE(qkv && out && grad_out && lse && grad_qkv && workspace, "ftx_attn_bwd: null pointer");
  FTX_REQUIRE(attn_tiling_built(qw, split), "ftx_attn_bwd_tiled: (%d, %d) is not a built tiling", qw, split);
  if (workspace_bytes < ftx_attn_bwd_workspace_bytes(b, t, h)) {
    set_error("ftx_attn_bwd: workspace %zu < required %zu", workspace_bytes, ftx_attn_bwd_workspace_bytes(b, t, h));
    return FTX_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float *delta = (float *)workspace;
  const int64_t rows = (int64_t)b * t * h;
  attn_delta_kernel<<<(unsigned)ceil_div(rows * 16, 256), 256, 0, st>>>(out, grad_out, rows, t, h, delta);
  attn_config(b, t, h, true, qw, split);
  ATTN_DISPATCH(attn_bwd_kv_kernel, qkv, grad_out, lse, delta, t, h, scale, grad_qkv);
  ATTN_DISPATCH(attn_bwd_q_kernel, qkv, grad_out, lse, delta, t, h, scale, grad_qkv);
  return check_launch("ftx_attn_bwd");
}

extern "C" int ftx_attn_bwd(const float *qkv, const float *out, const float *grad_out, const float *lse, int32_t b, int32_t t, int32_t h,
                            int32_t d, float scale, float *grad_qkv, void *workspace, size_t workspace_bytes, void *stream) {
  return ftx_attn_bwd_tiled(qkv, out, grad_out, lse, b, t, h, d, scale, grad_qkv, workspace, workspace_bytes, 0, 0, stream);
}
