// Sparse convolution on gfx950: output-stationary implicit GEMM over the kernel
// offsets with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
//   out[r,:] = sum_k A[tbl[k,r],:] @ Wk          (ftx_spconv_gemm)
//   dW[k]    = sum_r A[tbl[k,r],:]^T @ G[r,:]    (ftx_spconv_wgrad)
//
// Every output row is owned by exactly one wave and written once: no float atomics,
// bit-reproducible.  The neighbour table is (kvol, n_out) so a wave's 32 rows read 128
// contiguous bytes of it per offset; the gathered rows of A are staged in LDS with 16-byte
// loads (8 lanes cover one 128-byte row piece).
//
// MFMA operand maps (cdna_hip_programming.md §3): A operand lane l holds A[i=l&31][k=l>>5],
// B operand lane l holds B[k=l>>5][j=l&31]; accumulator reg g of lane l is
// C[row=(g&3)+8*(g>>2)+4*(l>>5)][col=l&31].  The reduction index is free to be permuted as
// long as A and B agree, so lane half h consumes k = 8t+4h+s for MFMA s of group t: the A
// fragment for four MFMAs is then ONE ds_read_b128 (row stride 36 floats: conflict-free).
#include "ftx_common.h"

using namespace ftx;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;        // reduction chunk (channels of A) staged per step
constexpr int AS_STRIDE = 36; // floats; 16-byte aligned rows, conflict-free ds_read_b128

template <int WM, int WN, int NT>
__global__ __launch_bounds__(64 * WM * WN) void spconv_gemm_kernel(const float *__restrict__ A, int64_t rows_a,
                                                                     const float *__restrict__ W, const int32_t *__restrict__ tbl,
                                                                     int64_t n_out, int ca, int co, int kvol, int w_transposed,
                                                                     float *__restrict__ out) {
  constexpr int NTHREADS = 64 * WM * WN;
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * NT * WN;
  constexpr int BS_STRIDE = BN + 4;
  constexpr int ROWS_PER_PASS = NTHREADS / 8;
  constexpr int A_PASSES = (BM + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr int B_VEC = BK * BN / 4;
  constexpr int B_PASSES = (B_VEC + NTHREADS - 1) / NTHREADS;

  __shared__ __attribute__((aligned(16))) float As[BM * AS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[BK * BS_STRIDE];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave % WM, wn = wave / WM;
  const int half = lane >> 5, l31 = lane & 31;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;

  const int arow = tid >> 3;        // row (within a pass) this thread gathers
  const int acol = (tid & 7) * 4;   // first of its 4 channels within the BK chunk

  for (int k = 0; k < kvol; ++k) {
    int32_t src[A_PASSES];
    int any = 0;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      int r = p * ROWS_PER_PASS + arow;
      int64_t gr = row0 + r;
      int32_t s = (r < BM && gr < n_out) ? tbl[(int64_t)k * n_out + gr] : -1;
      if (s >= rows_a) s = -1;
      src[p] = s;
      any |= (s >= 0);
    }
    if (!__syncthreads_or(any)) continue;  // no row of this tile has a neighbour at offset k

    const float *Wk = W + (int64_t)k * ca * co;
    for (int c0 = 0; c0 < ca; c0 += BK) {
      // ---- stage gathered A rows (zeros for absent neighbours / channel tail)
#pragma unroll
      for (int p = 0; p < A_PASSES; ++p) {
        int r = p * ROWS_PER_PASS + arow;
        if (r < BM) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (src[p] >= 0 && c0 + acol < ca) v = *(const float4 *)&A[(int64_t)src[p] * ca + c0 + acol];
          *(float4 *)&As[r * AS_STRIDE + acol] = v;
        }
      }
      // ---- stage the W[k] chunk: Bs[kk][n] = Wk[c0+kk][n0+n]
      if (!w_transposed) {
#pragma unroll
        for (int q = 0; q < B_PASSES; ++q) {
          int e = q * NTHREADS + tid;
          if (e < B_VEC) {
            int kk = e / (BN / 4);
            int n4 = (e - kk * (BN / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + kk < ca && n0 + n4 < co) v = *(const float4 *)&Wk[(int64_t)(c0 + kk) * co + n0 + n4];
            *(float4 *)&Bs[kk * BS_STRIDE + n4] = v;
          }
        }
      } else {
        // W[k] stored (co, ca): read 16 bytes along ca, scatter the 4 values down a Bs column
#pragma unroll
        for (int q = 0; q < B_PASSES; ++q) {
          int e = q * NTHREADS + tid;
          if (e < B_VEC) {
            int nn = e / (BK / 4);
            int k4 = (e - nn * (BK / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + nn < co && c0 + k4 < ca) v = *(const float4 *)&Wk[(int64_t)(n0 + nn) * ca + c0 + k4];
            Bs[(k4 + 0) * BS_STRIDE + nn] = v.x;
            Bs[(k4 + 1) * BS_STRIDE + nn] = v.y;
            Bs[(k4 + 2) * BS_STRIDE + nn] = v.z;
            Bs[(k4 + 3) * BS_STRIDE + nn] = v.w;
          }
        }
      }
      __syncthreads();
      // ---- MFMA over the chunk
      const float *arow_p = &As[(wm * 32 + l31) * AS_STRIDE + 4 * half];
      const float *bcol_p = &Bs[(4 * half) * BS_STRIDE + wn * 32 * NT + l31];
#pragma unroll
      for (int t = 0; t < BK / 8; ++t) {
        float4 a = *(const float4 *)(arow_p + 8 * t);
        float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            float b = bcol_p[(8 * t + s) * BS_STRIDE + j * 32];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b, acc[j], 0, 0, 0);
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- write the tile (each 32-lane half stores 128 contiguous bytes per register)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    int col = n0 + wn * 32 * NT + j * 32 + l31;
    if (col < co) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int64_t r = row0 + wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * half;
        if (r < n_out) out[r * co + col] = acc[j][g];
      }
    }
  }
}

template <int WM, int WN, int NT>
static void launch_gemm(const float *A, int64_t rows_a, const float *W, const int32_t *tbl, int64_t n_out, int ca, int co, int kvol,
                        int wT, float *out, hipStream_t st) {
  constexpr int BM = 32 * WM, BN = 32 * NT * WN;
  dim3 grid((unsigned)ceil_div(n_out, BM), (unsigned)ceil_div(co, BN));
  spconv_gemm_kernel<WM, WN, NT><<<grid, 64 * WM * WN, 0, st>>>(A, rows_a, W, tbl, n_out, ca, co, kvol, wT, out);
}

extern "C" int ftx_spconv_gemm(const float *A, int64_t rows_a, const float *W, const int32_t *tbl, int64_t n_out, int32_t ca, int32_t co,
                               int32_t kvol, int32_t w_transposed, float *out, void *stream) {
  FTX_REQUIRE(n_out >= 0 && rows_a >= 0 && kvol >= 1, "ftx_spconv_gemm: bad size");
  FTX_REQUIRE(ca >= 4 && ca % 4 == 0 && co >= 4 && co % 4 == 0, "ftx_spconv_gemm: channels must be multiples of 4 (ca=%d co=%d)", ca, co);
  if (n_out == 0) return FTX_OK;
  FTX_REQUIRE(W && tbl && out && (A || rows_a == 0), "ftx_spconv_gemm: null pointer");
  hipStream_t st = (hipStream_t)stream;
  // Tall tiles (128 rows) when the level has enough rows to fill the chip, otherwise
  // 32-row tiles with the 4 waves spread over the output channels.
  const int64_t tall_blocks = ceil_div(n_out, 128) * ceil_div(co, 128);
  const bool tall = tall_blocks >= 256 || co <= 32;
  if (tall) {
    int nt = co >= 128 ? 4 : (co + 31) / 32;
    if (co > 128 && co % 96 == 0 && co % 128 != 0) nt = 3;
    switch (nt) {
      case 1: launch_gemm<4, 1, 1>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st); break;
      case 2: launch_gemm<4, 1, 2>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st); break;
      case 3: launch_gemm<4, 1, 3>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st); break;
      default: launch_gemm<4, 1, 4>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st); break;
    }
  } else {
    if (co <= 128)
      launch_gemm<1, 4, 1>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st);
    else if (co % 96 == 0 && co % 128 != 0)
      launch_gemm<1, 4, 3>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st);
    else
      launch_gemm<1, 4, 2>(A, rows_a, W, tbl, n_out, ca, co, kvol, w_transposed, out, st);
  }
  return check_launch("ftx_spconv_gemm");
}

// ---------------------------------------------------------------------------------------
// Weight gradient.  Block (k, chunk, mt, nt) reduces the rows of its chunk for offset k into a
// 128(ca) x 128(cg) tile of dW[k]; rows of the chunk WITH a neighbour at offset k are first
// compacted (ordered, so the sum order is fixed) so the MFMAs only see real pairs.
// Chunks are combined by a second, deterministic pass.
// ---------------------------------------------------------------------------------------
constexpr int WG_CHUNK = 1024;   // rows compacted per round
constexpr int WG_BR = 32;        // compacted pairs staged per MFMA step
constexpr int WG_TM = 128, WG_TN = 128;
constexpr int WG_STRIDE = WG_TM + 4;

__global__ __launch_bounds__(256) void spconv_wgrad_kernel(const float *__restrict__ A, int64_t rows_a, const float *__restrict__ G,
                                                            const int32_t *__restrict__ tbl, int64_t n_rows, int ca, int cg, int kvol,
                                                            int64_t rows_per_chunk, int nchunks, float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float As[WG_BR * WG_STRIDE];
  __shared__ __attribute__((aligned(16))) float Gs[WG_BR * WG_STRIDE];
  __shared__ int32_t pair_a[WG_CHUNK];
  __shared__ int32_t pair_r[WG_CHUNK];
  __shared__ int wave_cnt[4];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int k = blockIdx.x % kvol;
  const int chunk = blockIdx.x / kvol;
  const int m0 = blockIdx.y * WG_TM, n0 = blockIdx.z * WG_TN;
  const int64_t r_begin = (int64_t)chunk * rows_per_chunk;
  const int64_t r_end = (r_begin + rows_per_chunk < n_rows) ? r_begin + rows_per_chunk : n_rows;

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;

  for (int64_t rb = r_begin; rb < r_end; rb += WG_CHUNK) {
    // ---- ordered compaction of the rows of this round that have a neighbour at offset k
    int npairs = 0;
    for (int sub = 0; sub < WG_CHUNK; sub += 256) {
      int64_t r = rb + sub + tid;
      int32_t s = (r < r_end) ? tbl[(int64_t)k * n_rows + r] : -1;
      if (s >= rows_a) s = -1;
      unsigned long long bal = __ballot(s >= 0);
      if (lane == 0) wave_cnt[wave] = __popcll(bal);
      __syncthreads();
      int base = npairs;
      for (int w = 0; w < wave; ++w) base += wave_cnt[w];
      int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      if (s >= 0) {
        int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        pair_a[pos] = s;
        pair_r[pos] = (int32_t)(r - rb);
      }
      npairs += total;
      __syncthreads();
    }
    // ---- MFMA over the compacted pairs, WG_BR at a time
    for (int p0 = 0; p0 < npairs; p0 += WG_BR) {
      // stage: 32 pairs x 128 channels for each operand = 1024 float4 each -> 4 per thread
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int e = q * 256 + tid;
        int pr = e >> 5;             // pair within the step
        int c4 = (e & 31) * 4;       // channel within the tile
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vg = va;
        if (p0 + pr < npairs) {
          if (m0 + c4 < ca) va = *(const float4 *)&A[(int64_t)pair_a[p0 + pr] * ca + m0 + c4];
          if (n0 + c4 < cg) vg = *(const float4 *)&G[(rb + pair_r[p0 + pr]) * cg + n0 + c4];
        }
        *(float4 *)&As[pr * WG_STRIDE + c4] = va;
        *(float4 *)&Gs[pr * WG_STRIDE + c4] = vg;
      }
      __syncthreads();
      // wave w owns channels [32w, 32w+32) of A (M) and all 128 of G (N)
#pragma unroll
      for (int s2 = 0; s2 < WG_BR / 2; ++s2) {
        int kk = 2 * s2 + half;
        float a = As[kk * WG_STRIDE + wave * 32 + l31];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float b = Gs[kk * WG_STRIDE + j * 32 + l31];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- partial tile -> part[chunk][k][ca][cg]
  float *dst = part + ((int64_t)chunk * kvol + k) * ca * cg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int col = n0 + j * 32 + l31;
    if (col < cg) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int row = m0 + wave * 32 + (g & 3) + 8 * (g >> 2) + 4 * half;
        if (row < ca) dst[(int64_t)row * cg + col] = acc[j][g];
      }
    }
  }
}

__global__ void wgrad_reduce_kernel(const float *__restrict__ part, int64_t elems, int nchunks, float *__restrict__ dW) {
  for (int64_t e = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; e < elems; e += (int64_t)gridDim.x * blockDim.x * 4) {
    float4 s = *(const float4 *)&part[e];
    for (int c = 1; c < nchunks; ++c) {
      float4 v = *(const float4 *)&part[(int64_t)c * elems + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *(float4 *)&dW[e] = s;
  }
}

static int wgrad_chunks(int64_t n_rows, int ca, int cg, int kvol) {
  int64_t tiles = (int64_t)kvol * ceil_div(ca, WG_TM) * ceil_div(cg, WG_TN);
  int64_t want = ceil_div(1024, tiles);                 // aim for ~1024 blocks
  int64_t max_chunks = ceil_div(n_rows, WG_CHUNK);      // at least one compaction round each
  if (want > max_chunks) want = max_chunks;
  if (want < 1) want = 1;
  return (int)want;
}

extern "C" size_t ftx_spconv_wgrad_workspace_bytes(int64_t n_rows, int32_t ca, int32_t cg, int32_t kvol) {
  if (n_rows <= 0 || ca <= 0 || cg <= 0 || kvol <= 0) return 256;
  int nchunks = wgrad_chunks(n_rows, ca, cg, kvol);
  if (nchunks <= 1) return 256;
  return sizeof(float) * (size_t)nchunks * kvol * ca * cg;
}

extern "C" int ftx_spconv_wgrad(const float *A, int64_t rows_a, const float *G, const int32_t *tbl, int64_t n_rows, int32_t ca,
                                int32_t cg, int32_t kvol, float *dW, void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n_rows >= 0 && rows_a >= 0 && kvol >= 1, "ftx_spconv_wgrad: bad size");
  FTX_REQUIRE(ca >= 4 && ca % 4 == 0 && cg >= 4 && cg % 4 == 0, "ftx_spconv_wgrad: channels must be multiples of 4 (ca=%d cg=%d)", ca, cg);
  FTX_REQUIRE(dW, "ftx_spconv_wgrad: null dW");
  hipStream_t st = (hipStream_t)stream;
  const int64_t elems = (int64_t)kvol * ca * cg;
  if (n_rows == 0) {
    if (hipMemsetAsync(dW, 0, sizeof(float) * elems, st) != hipSuccess) return check_launch("ftx_spconv_wgrad memset");
    return FTX_OK;
  }
  FTX_REQUIRE(A && G && tbl, "ftx_spconv_wgrad: null pointer");
  const int nchunks = wgrad_chunks(n_rows, ca, cg, kvol);
  int64_t rows_per_chunk = ceil_div(ceil_div(n_rows, nchunks), WG_CHUNK) * WG_CHUNK;
  float *part = dW;
  if (nchunks > 1) {
    size_t need = sizeof(float) * (size_t)nchunks * elems;
    if (!workspace || workspace_bytes < need) {
      set_error("ftx_spconv_wgrad: workspace %zu < required %zu", workspace_bytes, need);
      return FTX_EWORKSPACE;
    }
    part = (float *)workspace;
  }
  dim3 grid((unsigned)(kvol * nchunks), (unsigned)ceil_div(ca, WG_TM), (unsigned)ceil_div(cg, WG_TN));
  spconv_wgrad_kernel<<<grid, 256, 0, st>>>(A, rows_a, G, tbl, n_rows, ca, cg, kvol, rows_per_chunk, nchunks, part);
  if (nchunks > 1) wgrad_reduce_kernel<<<grid_for(elems / 4, 256), 256, 0, st>>>(part, elems, nchunks, dW);
  return check_launch("ftx_spconv_wgrad");
}
