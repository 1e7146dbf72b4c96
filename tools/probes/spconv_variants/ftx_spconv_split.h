// Pair GEMM of the sparse convolution on the bf16 matrix cores with f32-equivalent accuracy (included by ftx_spconv.hip).
//
// OPT-IN (ftx_spconv_set_split(1) / FTX_SPCONV_SPLIT=1); the default path stays the exact-f32 MFMA kernel above it.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  Every f32 value is the EXACT sum of three bf16 values
// (a = a1 + a2 + a3: round a to bf16, subtract -- exact --, round the remainder, subtract, round: 3 x 8 significand bits = the 24 of an
// f32), every bf16 x bf16 product is exact in f32, and the six products a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1 differ from a b by the
// three dropped ones, at most 2^-26 |a b| (|a2| <= 2^-9 |a|, |a3| <= 2^-18 |a|): below the 2^-24 rounding of one f32 multiply.  So
//     acc += sum over the six (plane_i of W, plane_j of A) pairs of  v_mfma_f32_32x32x16_bf16
// accumulates, in f32, products that are at least as accurate as the f32 MFMA's, in 6 x 32 cycles per 16 reduction steps instead
// of 8 x 64: 2.67x less matrix-pipe time.  Results are deterministic (fixed order) but not bit-identical to the f32-MFMA kernel
// (different summation tree inside the bf16 MFMA).
//
// Same tile as pairs_gemm_kernel (128 pairs of one offset x <= 128 output channels, 4 waves x 32 pairs, BK = 32), same tile
// search, gathers and 16-byte epilogue.  What changes is the staging: the gathered f32 chunk is split in registers and stored as
// three bf16 planes [row][32 k] (64-byte rows, 16-byte chunk index XOR-ed with (row >> 2) & 3: conflict-free ds_write_b64 and
// ds_read_b128), and a lane's fragment of a 16-deep reduction block is ONE ds_read_b128 per plane.
#pragma once

namespace split {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int ROWB = 64;   // bytes per LDS row: 32 bf16

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); }

// x = h + m + l exactly, each a bf16 (round-to-nearest-even at every step)
__device__ __forceinline__ void split1(float x, __bf16 &h, __bf16 &m, __bf16 &l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

__device__ __forceinline__ void split4(const float4 v, bf16x4 &h, bf16x4 &m, bf16x4 &l) {
  __bf16 a, b, c;
  split1(v.x, a, b, c); h[0] = a; m[0] = b; l[0] = c;
  split1(v.y, a, b, c); h[1] = a; m[1] = b; l[1] = c;
  split1(v.z, a, b, c); h[2] = a; m[2] = b; l[2] = c;
  split1(v.w, a, b, c); h[3] = a; m[3] = b; l[3] = c;
}

template <int NT, int WT>   // WT = 1: rows of W[k] are reduction-contiguous (the data-gradient call); 0: reduction-strided (forward)
__global__ __launch_bounds__(256) void pairs_gemm_split_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ gather,
                                                               const float *__restrict__ W, const int32_t *__restrict__ koff,
                                                               int ca, int co, int kvol, float *__restrict__ tmp, const float *__restrict__ bias,
                                                               int64_t n_dense, const int32_t *__restrict__ scatter, int64_t rows_out) {
  constexpr int TILE = TILE_P;
  constexpr int BN = 32 * NT;
  constexpr int A_PLANE = TILE * ROWB, B_PLANE = BN * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * A_PLANE + 3 * B_PLANE];
  __shared__ int s_tile[3];
  unsigned char *As = lds, *Bs = lds + 3 * A_PLANE;

  const int tid = threadIdx.x;
  if (gather == nullptr) {
    if (tid == 0) {
      int64_t left = n_dense - (int64_t)blockIdx.x * TILE;
      s_tile[0] = left > 0 ? 0 : -1;
      s_tile[1] = blockIdx.x * TILE;
      s_tile[2] = left > TILE ? TILE : (int)left;
    }
  } else if (tid < 64) {
    const int lane0 = tid;
    const int b = blockIdx.x;
    int c = (lane0 < kvol) ? koff[lane0 + 1] - koff[lane0] : 0;
    int nt = (c + TILE - 1) / TILE;
    int incl = nt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, 64);
      if (lane0 >= off) incl += v;
    }
    int excl = incl - nt;
    bool mine = (lane0 < kvol) && b >= excl && b < incl;
    unsigned long long m = __ballot(mine);
    if (mine) {
      int t = b - excl;
      int left = c - t * TILE;
      s_tile[0] = lane0;
      s_tile[1] = koff[lane0] + t * TILE;
      s_tile[2] = left > TILE ? TILE : left;
    }
    if (m == 0ull && lane0 == 0) s_tile[0] = -1;
  }
  __syncthreads();
  const int k = s_tile[0];
  if (k < 0) return;
  const int p0 = s_tile[1], cnt = s_tile[2];

  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.y * BN;
  const int arow = tid >> 3, acol = (tid & 7) * 4;
  const bool kfull = (ca % BK) == 0;

  int32_t src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int r = p * 32 + arow;
    int32_t s = 0;
    if (r < cnt) s = gather ? gather[p0 + r] : p0 + r;
    if (s < 0 || s >= rows_a) s = 0;
    src[p] = s;
  }
  const float *Wk = W + (int64_t)k * ca * co;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;

  // register stage of the next chunk: A 4 x float4 (4 consecutive k of one gathered row); W either NT x float4 along k (w_transposed:
  // rows of W are reduction-contiguous) or UNITS x 8 scalars (k-strided W: one unit = 8 consecutive k of one output channel)
  constexpr int UNITS = (BN * 4 + 255) / 256;
  float4 ra[4], rbt[WT ? NT : 1];
  float rbs[WT ? 1 : UNITS][8];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kfull || c0 + acol < ca) v = *(const float4 *)&A[(int64_t)src[p] * ca + c0 + acol];
      ra[p] = v;
    }
    if (WT) {
#pragma unroll
      for (int q = 0; q < (WT ? NT : 1); ++q) {
        int e = q * 256 + tid;
        int nn = n0 + (e >> 3), k4 = (e & 7) * 4;
        nn = nn < co ? nn : co - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kfull || c0 + k4 < ca) v = *(const float4 *)&Wk[(int64_t)nn * ca + c0 + k4];
        rbt[q] = v;
      }
    } else {
#pragma unroll
      for (int u = 0; u < (WT ? 1 : UNITS); ++u) {
        int e = u * 256 + tid;
        int nn = e % BN, ch = e / BN;
        int col = n0 + nn;
        col = col < co ? col : co - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          int kk = c0 + ch * 8 + i;
          float v = 0.f;
          if (ch < 4 && (kfull || kk < ca)) v = Wk[(int64_t)kk * co + col];
          rbs[u][i] = v;
        }
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      bf16x4 h, m, l;
      split4(ra[p], h, m, l);
      const int off = lds_off(p * 32 + arow, (tid & 7) >> 1) + (tid & 1) * 8;
      *(bf16x4 *)(As + off) = h;
      *(bf16x4 *)(As + A_PLANE + off) = m;
      *(bf16x4 *)(As + 2 * A_PLANE + off) = l;
    }
    if (WT) {
#pragma unroll
      for (int q = 0; q < (WT ? NT : 1); ++q) {
        bf16x4 h, m, l;
        split4(rbt[q], h, m, l);
        const int e = q * 256 + tid;
        const int off = lds_off(e >> 3, (e & 7) >> 1) + (e & 1) * 8;
        *(bf16x4 *)(Bs + off) = h;
        *(bf16x4 *)(Bs + B_PLANE + off) = m;
        *(bf16x4 *)(Bs + 2 * B_PLANE + off) = l;
      }
    } else {
#pragma unroll
      for (int u = 0; u < (WT ? 1 : UNITS); ++u) {
        const int e = u * 256 + tid;
        const int nn = e % BN, ch = e / BN;
        if (ch < 4) {
          bf16x8 h, m, l;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __bf16 a, b, c;
            split1(rbs[u][i], a, b, c);
            h[i] = a; m[i] = b; l[i] = c;
          }
          const int off = lds_off(nn, ch);
          *(bf16x8 *)(Bs + off) = h;
          *(bf16x8 *)(Bs + B_PLANE + off) = m;
          *(bf16x8 *)(Bs + 2 * B_PLANE + off) = l;
        }
      }
    }
  };

  load_chunk(0);
  for (int c0 = 0; c0 < ca; c0 += BK) {
    store_chunk();
    __syncthreads();
    if (c0 + BK < ca) load_chunk(c0 + BK);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int ch = 2 * kb + half;      // this lane's 8 reduction steps of the 16-deep block: k = 16 kb + 8 half + j
      const int aoff = lds_off(wave * 32 + l31, ch);
      const bf16x8 a0 = *(const bf16x8 *)(As + aoff), a1 = *(const bf16x8 *)(As + A_PLANE + aoff), a2 = *(const bf16x8 *)(As + 2 * A_PLANE + aoff);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int boff = lds_off(j * 32 + l31, ch);
        const bf16x8 b0 = *(const bf16x8 *)(Bs + boff), b1 = *(const bf16x8 *)(Bs + B_PLANE + boff), b2 = *(const bf16x8 *)(Bs + 2 * B_PLANE + boff);
        // smallest terms first; W planes are the MFMA's row operand (accumulator = (channel, pair), as in the f32 kernel)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a0, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a2, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const bool nfull = n0 + BN <= co;
  const int row = wave * 32 + l31;
  int64_t drow = row < cnt ? p0 + row : -1;
  bool zero = false;
  if (gather != nullptr && drow >= 0) {
    const int32_t sidx = gather[drow];
    zero = sidx < 0 || sidx >= rows_a;
  }
  if (scatter != nullptr && drow >= 0) {
    drow = scatter[drow];
    if (drow >= rows_out) drow = -1;
  }
  if (drow >= 0) {
    float *dst = tmp + drow * co;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + j * 32 + 8 * q + 4 * half;
        if (nfull || col < co) {
          float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
          if (zero) v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias) {
            const float4 bv = *(const float4 *)&bias[col];
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          *(float4 *)&dst[col] = v;
        }
      }
  }
}

template <int WT>
static void launch_wt(int nt, dim3 grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W,
                      const int32_t *koff, int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense, const int32_t *scatter,
                      int64_t rows_out) {
  switch (nt) {
    case 1: pairs_gemm_split_kernel<1, WT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
    case 2: pairs_gemm_split_kernel<2, WT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
    case 3: pairs_gemm_split_kernel<3, WT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
    default: pairs_gemm_split_kernel<4, WT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
  }
}

static void launch(int nt, dim3 grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W, int wT,
                   const int32_t *koff, int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense, const int32_t *scatter,
                   int64_t rows_out) {
  if (wT)
    launch_wt<1>(nt, grid, st, A, rows_a, gather, W, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out);
  else
    launch_wt<0>(nt, grid, st, A, rows_a, gather, W, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out);
}

}  // namespace split
