// Pair GEMM with LDS-DMA staging (included by ftx_spconv.hip): the same tile as pairs_gemm_kernel -- 128 pairs of
// ONE kernel offset x BN = 32 NT output channels, 4 waves x 32 pairs, exact-fp32 MFMA with W as the row operand --
// but the operands of a 32-deep chunk go global -> LDS directly (`global_load_lds_dwordx4`): no staging registers, no
// ds_write pass, one `s_waitcnt vmcnt(0)` + one barrier per step, the next chunk in flight under the MFMAs of the
// current one.  tools/probes/mfma_loop.hip measured this structure at 88 TFLOP/s on a synthetic level-0 128 -> 96 tile
// stream; in the library, with the tile search and the real maps, it ties the register-staged kernel (see the note at
// gemm_use_dma in ftx_spconv.hip), so it ships as an opt-in alternative.
//
// A DMA writes 64 lanes x 16 B contiguously, so LDS rows cannot be padded; bank conflicts are avoided by a swizzle
// applied to the per-lane SOURCE address and again by the reader:
//   A image [128 pairs][8 x 16 B]: the 16-byte piece c of row r is stored at position c ^ ((r >> 1) & 7)
//   W image, W[k] stored (co, ca) (dgrad):   [BN][8 x 16 B], same swizzle, fragments by ds_read_b128
//   W image, W[k] stored (ca, co) (forward): [32][BN] floats as they come, fragments by ds_read_b32
// Whole chunks only: ca % 32 == 0 and co % BN == 0 (everything else stays on pairs_gemm_kernel).  Rows past a
// tile's end gather some valid row and are masked at the store.
#pragma once

namespace dma {

constexpr int TILE = 128;
constexpr int BK = 32;
constexpr int A_BYTES = TILE * BK * 4;

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  // M0 carries the wave-uniform LDS destination; each lane's 16 bytes land at lds_dst + lane * 16
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int NT>
__global__ __launch_bounds__(256) void pairs_gemm_dma_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ gather,
                                                             const float *__restrict__ W, int w_transposed, const int32_t *__restrict__ koff,
                                                             int ca, int co, int kvol, float *__restrict__ tmp, const float *__restrict__ bias,
                                                             int64_t n_dense) {
  constexpr int BN = 32 * NT;
  constexpr int W_BYTES = BN * BK * 4;
  constexpr int STAGE = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [2 stages][A image | W image]
  __shared__ int s_tile[3];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

  const int tid = threadIdx.x;
  if (gather == nullptr) {
    if (tid == 0) {
      int64_t left = n_dense - (int64_t)blockIdx.x * TILE;
      s_tile[0] = left > 0 ? 0 : -1;
      s_tile[1] = blockIdx.x * TILE;
      s_tile[2] = left > TILE ? TILE : (int)left;
    }
  } else if (tid < 64) {
    // tile -> (offset, first pair, pair count): wave 0 scans the per-offset tile counts
    const int b = blockIdx.x;
    int c = (tid < kvol) ? koff[tid + 1] - koff[tid] : 0;
    int nt = (c + TILE - 1) / TILE;
    int incl = nt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, 64);
      if (tid >= off) incl += v;
    }
    int excl = incl - nt;
    bool mine = (tid < kvol) && b >= excl && b < incl;
    unsigned long long m = __ballot(mine);
    if (mine) {
      int t = b - excl;
      int left = c - t * TILE;
      s_tile[0] = tid;
      s_tile[1] = koff[tid] + t * TILE;
      s_tile[2] = left > TILE ? TILE : left;
    }
    if (m == 0ull && tid == 0) s_tile[0] = -1;
  }
  __syncthreads();
  const int k = s_tile[0];
  if (k < 0) return;  // surplus block of the upper-bound grid
  const int p0 = s_tile[1], cnt = s_tile[2];

  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * BN;
  const float *Wk = W + (int64_t)k * ca * co;

  // this lane's 4 A pieces per chunk: rows wave*32 + u*8 + (lane >> 3), stored position lane & 7
  const float *a_src[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = wave * 32 + u * 8 + (lane >> 3);
    int64_t row = p0 + (r < cnt ? r : 0);             // rows past the tile's end: any valid row (masked at the store)
    if (gather) row = gather[row];
    if (row < 0 || row >= rows_a) row = 0;
    const int piece = (lane & 7) ^ ((r >> 1) & 7);    // logical 16-byte piece kept at this lane's position
    a_src[u] = A + row * ca + piece * 4;
  }
  // this lane's NT W pieces per chunk
  const float *w_src[NT];
  int w_step;                                          // floats to advance per 32-deep chunk
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int i = u * 4 + wave;                        // 1-KiB piece of the W image
    if (!w_transposed) {                               // image [32 k][BN]: W[k] rows are (ca, co)
      const int e = i * 64 + lane;                     // float4 index inside the image
      const int kk = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
      w_src[u] = Wk + (int64_t)kk * co + n0 + n4;
    } else {                                           // image [BN n][8 pieces], swizzled like A: W[k] rows are (co, ca)
      const int n = i * 8 + (lane >> 3);
      const int piece = (lane & 7) ^ ((n >> 1) & 7);
      w_src[u] = Wk + (int64_t)(n0 + n) * ca + piece * 4;
    }
  }
  w_step = w_transposed ? BK : BK * co;

  auto issue = [&](int step, int buf) {
    const unsigned sa = lds0 + buf * STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) glds16(a_src[u] + step * BK, sa + (wave * 32 + u * 8) * 128);
#pragma unroll
    for (int u = 0; u < NT; ++u) glds16(w_src[u] + (int64_t)step * w_step, sa + A_BYTES + (u * 4 + wave) * 1024);
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;

  // fragment addresses: row l31 of a 32-row group, piece (2t + half) ^ ((l31 >> 1) & 7)
  const int fsw = (l31 >> 1) & 7;
  int frag_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) frag_off[t] = l31 * 128 + (((2 * t + half) ^ fsw) << 4);

  const int steps = ca / BK;
  issue(0, 0);
  for (int it = 0; it < steps; ++it) {
    // the chunk of this step has landed for every wave; the other buffer (read last step) is free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + 1 < steps) issue(it + 1, (it + 1) & 1);

    const char *sa = smem + (it & 1) * STAGE;
    const char *sw = sa + A_BYTES;
    const char *ap = sa + wave * 32 * 128;
    float af[2][4], bf[2][NT][4];
    auto load_frag = [&](int buf, int t) {
      float4 a = *(const float4 *)(ap + frag_off[t]);
      af[buf][0] = a.x; af[buf][1] = a.y; af[buf][2] = a.z; af[buf][3] = a.w;
      if (w_transposed) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float4 b = *(const float4 *)(sw + j * 32 * 128 + frag_off[t]);
          bf[buf][j][0] = b.x; bf[buf][j][1] = b.y; bf[buf][j][2] = b.z; bf[buf][j][3] = b.w;
        }
      } else {
        const float *wp = (const float *)sw + (8 * t + 4 * half) * BN + l31;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int s = 0; s < 4; ++s) bf[buf][j][s] = wp[s * BN + j * 32];
      }
    };
    load_frag(0, 0);
#pragma unroll
    for (int t = 0; t < BK / 8; ++t) {
      if (t + 1 < BK / 8) load_frag((t + 1) & 1, t + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[t & 1][j][s], af[t & 1][s], acc[j], 0, 0, 0);
    }
  }

  // lane (pair l31, half) holds 4 consecutive output channels in every 4 consecutive accumulator registers
  const int row = wave * 32 + l31;
  if (row < cnt) {
    float *dst = tmp + (int64_t)(p0 + row) * co;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + j * 32 + 8 * q + 4 * half;
        float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
        if (bias) {
          const float4 bv = *(const float4 *)&bias[col];
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        *(float4 *)&dst[col] = v;
      }
  }
}

template <int NT>
static int launch(dim3 grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W, int wT, const int32_t *koff,
                  int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense) {
  constexpr int LDS = 2 * (A_BYTES + 32 * NT * BK * 4);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void *)pairs_gemm_dma_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    configured = true;
  }
  pairs_gemm_dma_kernel<NT><<<grid, 256, LDS, st>>>(A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense);
  return 0;
}

static int dispatch(int nt, dim3 grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W, int wT,
                    const int32_t *koff, int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense) {
  switch (nt) {
    case 1: return launch<1>(grid, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense);
    case 2: return launch<2>(grid, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense);
    case 3: return launch<3>(grid, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense);
    default: return launch<4>(grid, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense);
  }
}

}  // namespace dma
