// Sparse convolution on gfx950 as  pair-list gather-GEMM  +  ordered reduce.
//
// A kernel map is stored as ONE compacted pair list, sorted by (offset k, output row):
//   pair_in[p], pair_out[p]            rows of the input / output tensor joined by pair p
//   koff[k] .. koff[k+1]               the pairs of offset k
//   pos[k, o] / pos_t[k, i]            position of the pair (k,o) / (k,i) in the list, or -1
//
//   forward      tmp[p]  = in[pair_in[p]]   @ W[k(p)]        out[o] = sum_k tmp[pos[k,o]]
//   data grad    tmp[p]  = gout[pair_out[p]] @ W[k(p)]^T     gin[i] = sum_k tmp[pos_t[k,i]]
//   weight grad  dW[k]   = sum_{p in k} in[pair_in[p]]^T @ gout[pair_out[p]]
//
// Only real (in,out) pairs reach the matrix cores (an output-stationary implicit GEMM spends
// ~80% of its MFMAs on absent neighbours: a voxel has ~5-9 of 27), every tile of 128 pairs
// shares one W[k], and the reduce adds each row's <= 27 partial rows in fixed k order: no float
// atomics anywhere, results are bit-reproducible.  tmp costs one extra streamed write + read of
// P x co floats, which is cheaper than the atomic rate (1.3 TB/s) by ~4x.
//
// MFMA: exact-fp32 v_mfma_f32_32x32x2_f32.  Operand maps (cdna_hip_programming.md §3): A operand
// lane l holds A[i=l&31][k=l>>5], B operand lane l holds B[k=l>>5][j=l&31]; accumulator reg g of
// lane l is C[row=(g&3)+8*(g>>2)+4*(l>>5)][col=l&31].  The reduction index may be permuted as
// long as A and B agree: lane half h consumes k = 8t+4h+s for MFMA s of group t, so the A fragment
// of four MFMAs is ONE ds_read_b128 (row stride 36 floats -> conflict-free).
#include <cstring>
#include <cstdlib>
#include "ftx_common.h"
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

using namespace ftx;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------
// pair lists
// ---------------------------------------------------------------------------------------
struct IsValid {
  __host__ __device__ int32_t operator()(int32_t v) const { return v >= 0 ? 1 : 0; }
};

extern "C" size_t ftx_kernel_map_count_workspace_bytes(int64_t n_out, int32_t k) {
  if (n_out <= 0 || k <= 0) return 256;
  size_t bytes = 0;
  const int32_t *in = nullptr;
  int32_t *out = nullptr;
  auto it = rocprim::make_transform_iterator(in, IsValid());
  if (rocprim::exclusive_scan(nullptr, bytes, it, out, 0, (size_t)(n_out * k), rocprim::plus<int32_t>()) != hipSuccess) return 0;
  return bytes + 256;
}

__global__ void koff_kernel(const int32_t *__restrict__ nbr, const int32_t *__restrict__ scan, int64_t n_out, int k, int32_t *__restrict__ koff) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < k) koff[t] = scan[(int64_t)t * n_out];
  if (t == k) {
    int64_t last = (int64_t)k * n_out - 1;
    koff[k] = scan[last] + (nbr[last] >= 0 ? 1 : 0);
  }
}

extern "C" int ftx_kernel_map_count(const int32_t *nbr, int64_t n_out, int32_t k, int32_t *pos, int32_t *koff, void *workspace,
                                    size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n_out >= 0 && k >= 1, "ftx_kernel_map_count: bad size");
  FTX_REQUIRE(koff, "ftx_kernel_map_count: null koff");
  hipStream_t st = (hipStream_t)stream;
  if (n_out == 0) {
    if (hipMemsetAsync(koff, 0, sizeof(int32_t) * (k + 1), st) != hipSuccess) return check_launch("ftx_kernel_map_count memset");
    return FTX_OK;
  }
  FTX_REQUIRE(nbr && pos && workspace, "ftx_kernel_map_count: null pointer");
  FTX_REQUIRE(n_out * k < 0x7fffffff, "ftx_kernel_map_count: map too large for int32 positions");
  size_t need = ftx_kernel_map_count_workspace_bytes(n_out, k);
  if (workspace_bytes < need) {
    set_error("ftx_kernel_map_count: workspace %zu < required %zu", workspace_bytes, need);
    return FTX_EWORKSPACE;
  }
  size_t bytes = workspace_bytes;
  auto it = rocprim::make_transform_iterator(nbr, IsValid());
  // pos temporarily holds the exclusive scan of the validity flags (k-major = sorted by (k, row))
  if (rocprim::exclusive_scan(workspace, bytes, it, pos, 0, (size_t)(n_out * k), rocprim::plus<int32_t>(), st) != hipSuccess) {
    set_error("ftx_kernel_map_count: scan failed");
    return FTX_ELAUNCH;
  }
  koff_kernel<<<1, 64, 0, st>>>(nbr, pos, n_out, k, koff);
  return check_launch("ftx_kernel_map_count");
}

__global__ void fill_m1_kernel(int32_t *__restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = -1;
}

__global__ void pairs_scatter_kernel(const int32_t *__restrict__ nbr, int64_t n_out, int64_t n_in, int k, int32_t *__restrict__ pos,
                                     int32_t *__restrict__ pos_t, int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_out,
                                     int64_t cap) {
  const int64_t total = n_out * k;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int32_t i = nbr[e];
    int32_t p = pos[e];
    if (i >= 0 && i < n_in && p < cap) {
      int kk = (int)(e / n_out);
      int64_t o = e - (int64_t)kk * n_out;
      pair_in[p] = i;
      pair_out[p] = (int32_t)o;
      pos_t[(int64_t)kk * n_in + i] = p;
    } else {
      pos[e] = -1;
    }
  }
}

extern "C" int ftx_kernel_map_pairs(const int32_t *nbr, int64_t n_out, int64_t n_in, int32_t k, int32_t *pos, int32_t *pos_t, int32_t *pair_in,
                                    int32_t *pair_out, int64_t n_pairs, void *stream) {
  FTX_REQUIRE(n_out >= 0 && n_in >= 0 && k >= 1 && n_pairs >= 0, "ftx_kernel_map_pairs: bad size");
  hipStream_t st = (hipStream_t)stream;
  if (n_in > 0) {
    FTX_REQUIRE(pos_t, "ftx_kernel_map_pairs: null pos_t");
    fill_m1_kernel<<<grid_for(n_in * k, 256), 256, 0, st>>>(pos_t, n_in * k);
  }
  if (n_out == 0) return check_launch("ftx_kernel_map_pairs");
  FTX_REQUIRE(nbr && pos && (n_pairs == 0 || (pair_in && pair_out)), "ftx_kernel_map_pairs: null pointer");
  pairs_scatter_kernel<<<grid_for(n_out * k, 256), 256, 0, st>>>(nbr, n_out, n_in, k, pos, pos_t, pair_in, pair_out, n_pairs);
  return check_launch("ftx_kernel_map_pairs");
}

// ---------------------------------------------------------------------------------------
// phase 1: tmp[p,:] = A[gather[p],:] @ Wk(p)        (tiles of 128 pairs of one offset)
// ---------------------------------------------------------------------------------------
constexpr int BK = 32;         // reduction chunk staged per step
constexpr int AS_STRIDE = 36;  // floats
constexpr int TILE_P = 128;    // pairs per tile and row-tile count RT: tile = RT x (4 waves x 32 pairs)

// NT = 32-column tiles per wave, RT = 32-pair row tiles per wave.  RT = 2 (256-pair tiles) halves the
// W[k] traffic, the B-operand LDS reads and the barriers per MFMA; used when every offset has
// enough pairs that the padding of its last tile does not matter.
template <int NT, int RT>
__global__ __launch_bounds__(256) void pairs_gemm_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ gather,
                                                         const float *__restrict__ W, int w_transposed, const int32_t *__restrict__ koff,
                                                         int ca, int co, int kvol, float *__restrict__ tmp, const float *__restrict__ bias,
                                                         int64_t n_dense, const int32_t *__restrict__ scatter, int64_t rows_out) {
  // gather == nullptr: dense mode, tmp[r,:] = A[r,:] @ W (+ bias) for r < n_dense (kvol = 1)
  // scatter != nullptr: the result row of pair p goes to row scatter[p] of `tmp` (rows_out rows) instead of row p: for maps in
  // which every destination row receives exactly ONE pair the convolution is this one launch, with no tmp and no reduce pass
  constexpr int TILE = TILE_P * RT;
  constexpr int BN = 32 * NT;
  constexpr int BS_STRIDE = AS_STRIDE;         // W chunk kept as Bs[n][k]: the reduction index is contiguous for BOTH operands
  constexpr int B_PASSES = NT;                 // BK * BN / 4 float4 per W chunk = NT * 256
  constexpr int A_PASSES = 4 * RT;

  __shared__ __attribute__((aligned(16))) float As[TILE * AS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[BN * BS_STRIDE];
  __shared__ int s_tile[3];

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
  if (k < 0) return;  // surplus block of the upper-bound grid
  const int p0 = s_tile[1], cnt = s_tile[2];

  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.y * BN;
  const int arow = tid >> 3, acol = (tid & 7) * 4;

  // Gathers are unconditional loads from addresses that are always valid: rows past the end of the tile (and malformed
  // indices) read row 0 -- what they produce lands in accumulator rows the epilogue never stores (or zeroes) -- and a column
  // tile that sticks out of W re-reads W's last float4.  Only a reduction dimension that is not a multiple of BK (the 4-channel
  // stem) needs zero fill, on a uniform slow path.  (A branch per load kept every load behind its own compare.)
  const bool kfull = (ca % BK) == 0;
  int32_t src[A_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    int r = p * 32 + arow;
    int32_t s = 0;
    if (r < cnt) s = gather ? gather[p0 + r] : p0 + r;
    if (s < 0 || s >= rows_a) s = 0;
    src[p] = s;
  }
  const float *Wk = W + (int64_t)k * ca * co;

  f32x16 acc[RT][NT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[r][j][g] = 0.f;

  float4 ra[A_PASSES], rb[B_PASSES];
  auto load_chunk = [&](int c0) {
    if (kfull) {
#pragma unroll
      for (int p = 0; p < A_PASSES; ++p) ra[p] = *(const float4 *)&A[(int64_t)src[p] * ca + c0 + acol];
#pragma unroll
      for (int q = 0; q < B_PASSES; ++q) {
        if (!w_transposed) {  // W[k] stored (ca, co): 16 bytes along co; a wave covers 8 k-rows x 128 B
          int kk = ((tid >> 6) << 3) + (tid & 7), n4 = n0 + (q * 8 + ((tid >> 3) & 7)) * 4;
          n4 = n4 + 4 <= co ? n4 : co - 4;
          rb[q] = *(const float4 *)&Wk[(int64_t)(c0 + kk) * co + n4];
        } else {              // W[k] stored (co, ca): 16 bytes along ca
          int e = q * 256 + tid;
          int nn = n0 + (e >> 3), k4 = (e & 7) * 4;
          nn = nn < co ? nn : co - 1;
          rb[q] = *(const float4 *)&Wk[(int64_t)nn * ca + c0 + k4];
        }
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + acol < ca) v = *(const float4 *)&A[(int64_t)src[p] * ca + c0 + acol];
      ra[p] = v;
    }
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!w_transposed) {
        int kk = ((tid >> 6) << 3) + (tid & 7), n4 = (q * 8 + ((tid >> 3) & 7)) * 4;
        if (c0 + kk < ca && n0 + n4 < co) v = *(const float4 *)&Wk[(int64_t)(c0 + kk) * co + n0 + n4];
      } else {
        int e = q * 256 + tid;
        int nn = e >> 3, k4 = (e & 7) * 4;
        if (n0 + nn < co && c0 + k4 < ca) v = *(const float4 *)&Wk[(int64_t)(n0 + nn) * ca + c0 + k4];
      }
      rb[q] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) *(float4 *)&As[(p * 32 + arow) * AS_STRIDE + acol] = ra[p];
#pragma unroll
    for (int q = 0; q < B_PASSES; ++q) {
      if (!w_transposed) {  // transposing store; (n4i * 144 + kl) mod 64 banks: 2-way conflicts at worst
        int kk = ((tid >> 6) << 3) + (tid & 7), n4 = (q * 8 + ((tid >> 3) & 7)) * 4;
        Bs[(n4 + 0) * BS_STRIDE + kk] = rb[q].x;
        Bs[(n4 + 1) * BS_STRIDE + kk] = rb[q].y;
        Bs[(n4 + 2) * BS_STRIDE + kk] = rb[q].z;
        Bs[(n4 + 3) * BS_STRIDE + kk] = rb[q].w;
      } else {
        int e = q * 256 + tid;
        int nn = e >> 3, k4 = (e & 7) * 4;
        *(float4 *)&Bs[nn * BS_STRIDE + k4] = rb[q];
      }
    }
  };

  load_chunk(0);
  for (int c0 = 0; c0 < ca; c0 += BK) {
    store_chunk();
    __syncthreads();
    if (c0 + BK < ca) load_chunk(c0 + BK);  // next chunk's global loads fly under the MFMAs
    // lane (l31, half) owns k = 8t + 4*half + s of both operands: one ds_read_b128 per operand row feeds 4 MFMAs.
    // The fragments of step t+1 are in flight while the MFMAs of step t issue.
    const float *arow_p = &As[(wave * 32 * RT + l31) * AS_STRIDE + 4 * half];
    const float *brow_p = &Bs[l31 * BS_STRIDE + 4 * half];
    float af[2][RT][4], bf[2][NT][4];
    auto load_frag = [&](int buf, int t) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        float4 a = *(const float4 *)(arow_p + r * 32 * AS_STRIDE + 8 * t);
        af[buf][r][0] = a.x; af[buf][r][1] = a.y; af[buf][r][2] = a.z; af[buf][r][3] = a.w;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float4 b = *(const float4 *)(brow_p + j * 32 * BS_STRIDE + 8 * t);
        bf[buf][j][0] = b.x; bf[buf][j][1] = b.y; bf[buf][j][2] = b.z; bf[buf][j][3] = b.w;
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
#pragma unroll
          for (int r = 0; r < RT; ++r)
            acc[r][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[t & 1][j][s], af[t & 1][r][s], acc[r][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- W is the MFMA's row operand, so lane (pair l31, half) holds 4 consecutive output channels in every 4
  // consecutive accumulator registers: 16-byte stores, each pair row receives 32 contiguous bytes per instruction
  // (dword stores of the (pair, channel) orientation took 17k cycles per tile here, these take 6k)
  const bool nfull = n0 + BN <= co;
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int row = wave * 32 * RT + r * 32 + l31;
    int64_t drow = row < cnt ? p0 + row : -1;
    bool zero = false;   // a pair whose source index is out of range contributes a zero row, as if it gathered zeros
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
            float4 v = make_float4(acc[r][j][4 * q], acc[r][j][4 * q + 1], acc[r][j][4 * q + 2], acc[r][j][4 * q + 3]);
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
}

template <int RT>
static void launch_pairs_gemm(int nt, dim3 grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W, int wT,
                              const int32_t *koff, int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense,
                              const int32_t *scatter = nullptr, int64_t rows_out = 0) {
  switch (nt) {
    case 1: pairs_gemm_kernel<1, RT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
    case 2: pairs_gemm_kernel<2, RT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
    case 3: pairs_gemm_kernel<3, RT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
    default: pairs_gemm_kernel<4, RT><<<grid, 256, 0, st>>>(A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out); break;
  }
}

#include "ftx_lastblock.h"

// One pair-GEMM kernel ships.  The LDS-DMA, producer / consumer and bf16x3 variants of rounds 1-2 each tied or lost against it
// (DESIGN.md section 8); their sources live under tools/probes/spconv_variants/ and are not part of libftx.so.  Nothing in this file
// is selected by process-wide mutable state or by a device query: tile shapes and workspace sizes are functions of the arguments.
// 32-column tiles per block as a function of the arguments: 128 columns per block where there are enough pair tiles to fill the chip,
// 64 where there are not (the two deepest levels: 159-445 tiles -- 372 blocks of 128 columns took 42.0 us on the 256 -> 256 layer of
// level 16, 744 blocks of 64 take 34.5; with 472 blocks and more the wider tile wins by 2-5 %, it reads every gathered row once).
// A column split changes no sum: the results are the same bits either way.
static int gemm_nt(int co, int64_t row_tiles) {
  int nt = co >= 128 ? 4 : (co + 31) / 32;
  if (co > 128 && co % 96 == 0 && co % 128 != 0) nt = 3;
  if (nt == 4 && row_tiles * ceil_div(co, 128) <= 400) nt = 2;
  return nt;
}

// Measured on MI355X (profiles/r01_spconv_layer_micro.txt workload): 256-pair tiles (RT = 2) are 5-30 % SLOWER than 128-pair tiles on every
// layer -- the extra accumulators cut occupancy to 1-2 waves per SIMD and the kernel is latency-, not W-traffic-bound: RT = 1 everywhere.

extern "C" int ftx_spconv_pairs_gemm(const float *A, int64_t rows_a, const int32_t *gather, const float *W, int32_t w_transposed,
                                     const int32_t *koff, int64_t n_pairs, int32_t ca, int32_t co, int32_t kvol, float *tmp, void *stream) {
  FTX_REQUIRE(n_pairs >= 0 && rows_a >= 0 && kvol >= 1 && kvol <= 64, "ftx_spconv_pairs_gemm: bad size");
  FTX_REQUIRE(ca >= 4 && ca % 4 == 0 && co >= 4 && co % 4 == 0, "ftx_spconv_pairs_gemm: channels must be multiples of 4 (ca=%d co=%d)", ca, co);
  if (n_pairs == 0) return FTX_OK;
  FTX_REQUIRE(A && gather && W && koff && tmp && rows_a >= 1, "ftx_spconv_pairs_gemm: null pointer or empty operand");
  hipStream_t st = (hipStream_t)stream;
  const unsigned tiles_ub = (unsigned)(ceil_div(n_pairs, TILE_P) + kvol);  // sum_k ceil(cnt_k/tile) <= P/tile + kvol
  const int nt = gemm_nt(co, tiles_ub);
  dim3 grid(tiles_ub, (unsigned)ceil_div(co, 32 * nt));
  launch_pairs_gemm<1>(nt, grid, st, A, rows_a, gather, W, w_transposed, koff, ca, co, kvol, tmp, nullptr, 0);
  return check_launch("ftx_spconv_pairs_gemm");
}

// One-launch convolution for maps whose destination side is a bijection of the pair list: out[scatter[p],:] = A[gather[p],:] @ Wk(p).
// The strided 2^3 convolution joins every fine voxel to exactly one (coarse voxel, offset), so its data gradient and the
// transposed convolution built on the same map (models/spvcnn.py:38-50) write every fine row exactly once: no tmp, no reduce.
// The caller guarantees that `scatter` is injective (rows it does not name are left untouched).
extern "C" int ftx_spconv_pairs_gemm_scatter(const float *A, int64_t rows_a, const int32_t *gather, const int32_t *scatter, const float *W,
                                             int32_t w_transposed, const int32_t *koff, int64_t n_pairs, int32_t ca, int32_t co, int32_t kvol,
                                             float *out, int64_t rows_out, void *stream) {
  FTX_REQUIRE(n_pairs >= 0 && rows_a >= 0 && rows_out >= 0 && kvol >= 1 && kvol <= 64, "ftx_spconv_pairs_gemm_scatter: bad size");
  FTX_REQUIRE(ca >= 4 && ca % 4 == 0 && co >= 4 && co % 4 == 0, "ftx_spconv_pairs_gemm_scatter: channels must be multiples of 4 (ca=%d co=%d)", ca, co);
  if (n_pairs == 0) return FTX_OK;
  FTX_REQUIRE(A && gather && scatter && W && koff && out && rows_a >= 1, "ftx_spconv_pairs_gemm_scatter: null pointer or empty operand");
  hipStream_t st = (hipStream_t)stream;
  const int nt = gemm_nt(co, ceil_div(n_pairs, TILE_P) + kvol);
  dim3 grid((unsigned)(ceil_div(n_pairs, TILE_P) + kvol), (unsigned)ceil_div(co, 32 * nt));
  launch_pairs_gemm<1>(nt, grid, st, A, rows_a, gather, W, w_transposed, koff, ca, co, kvol, out, nullptr, 0, scatter, rows_out);
  return check_launch("ftx_spconv_pairs_gemm_scatter");
}

// Dense rows: out[r,:] = A[r,:] @ W (+ bias) on the same tile code (identity gather, one "offset").
// The point-branch Linear layers, the 1x1x1 convolutions and the heads are skinny GEMMs
// (81k rows x 20..256 columns, K = 32..256) that are HBM-bound: rows in, rows out, W from L2.
extern "C" int ftx_rows_gemm(const float *A, int64_t n, const float *W, int32_t w_transposed, const float *bias, int32_t ca, int32_t co,
                             float *out, void *stream) {
  FTX_REQUIRE(n >= 0, "ftx_rows_gemm: n < 0");
  FTX_REQUIRE(ca >= 4 && ca % 4 == 0 && co >= 4 && co % 4 == 0, "ftx_rows_gemm: channels must be multiples of 4 (ca=%d co=%d)", ca, co);
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(A && W && out, "ftx_rows_gemm: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int nt = gemm_nt(co, ceil_div(n, TILE_P));
  dim3 grid((unsigned)ceil_div(n, TILE_P), (unsigned)ceil_div(co, 32 * nt));
  launch_pairs_gemm<1>(nt, grid, st, A, n, nullptr, W, w_transposed, nullptr, ca, co, 1, out, bias, n);
  return check_launch("ftx_rows_gemm");
}

// ---------------------------------------------------------------------------------------
// phase 2: out[r,:] = sum_k tmp[pos[k,r],:]   (fixed k order; rows without pairs become 0)
// ---------------------------------------------------------------------------------------
template <int KVOL>
__global__ void spconv_reduce_kernel(const float *__restrict__ tmp, const int32_t *__restrict__ pos, int64_t n, int co, int kvol,
                                     float *__restrict__ out) {
  const int cv = co >> 2;
  const int64_t total = n * cv;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e / cv;
    int j = (int)(e - r * cv) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KVOL > 0) {
      int32_t p[KVOL > 0 ? KVOL : 1];
#pragma unroll
      for (int k = 0; k < KVOL; ++k) p[k] = pos[(int64_t)k * n + r];
#pragma unroll
      for (int k = 0; k < KVOL; ++k) {
        if (p[k] >= 0) {
          float4 v = *(const float4 *)&tmp[(int64_t)p[k] * co + j];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    } else {
      for (int k = 0; k < kvol; ++k) {
        int32_t q = pos[(int64_t)k * n + r];
        if (q >= 0) {
          float4 v = *(const float4 *)&tmp[(int64_t)q * co + j];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    }
    *(float4 *)&out[r * co + j] = acc;
  }
}

extern "C" int ftx_spconv_reduce(const float *tmp, const int32_t *pos, int64_t n, int32_t co, int32_t kvol, float *out, void *stream) {
  FTX_REQUIRE(n >= 0 && kvol >= 1 && co >= 4 && co % 4 == 0, "ftx_spconv_reduce: bad size");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(pos && out, "ftx_spconv_reduce: null pointer");
  int64_t work = n * (co / 4);
  int64_t g = ceil_div(work, 256);
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (kvol == 27)
    spconv_reduce_kernel<27><<<(unsigned)g, 256, 0, st>>>(tmp, pos, n, co, kvol, out);
  else if (kvol == 8)
    spconv_reduce_kernel<8><<<(unsigned)g, 256, 0, st>>>(tmp, pos, n, co, kvol, out);
  else
    spconv_reduce_kernel<0><<<(unsigned)g, 256, 0, st>>>(tmp, pos, n, co, kvol, out);
  return check_launch("ftx_spconv_reduce");
}

// The same reduce, also producing the BatchNorm statistics of its output (sum and sum of squares per channel, float64) as
// per-block partials: the BatchNorm that follows every convolution (models/spvcnn.py:22-35,53-79) then needs no pass of its own
// over `out`.  Block = (256 / (co/4)) rows x co/4 float4 columns over a contiguous row range; a thread keeps one column, so its
// eight float64 sums stay in registers; rows of a block are summed in a fixed order and blocks are combined in block order by
// ftx_lastblock.h (then one apply launch, ftx_bn_train_fwd_totals): the statistics do not depend on the launch geometry beyond `nb`, and are bit-reproducible.
template <int KVOL>
__global__ __launch_bounds__(256) void spconv_reduce_stats_kernel(const float *__restrict__ tmp, const int32_t *__restrict__ pos, int64_t n, int co,
                                                                  int kvol, float *__restrict__ out, double *part, StreamScratch sc) {
  extern __shared__ double sh[];  // [2][RL][co]
  const int cv = co >> 2;
  const int RL = 256 / cv;
  const int tid = threadIdx.x;
  const int cg = tid % cv, rl = tid / cv;
  const int64_t rows_per_block = ceil_div(n, (int64_t)gridDim.x);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < n) ? r0 + rows_per_block : n;
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if (rl < RL) {
    const int j = cg * 4;
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KVOL > 0) {
        int32_t p[KVOL > 0 ? KVOL : 1];
#pragma unroll
        for (int k = 0; k < KVOL; ++k) p[k] = pos[(int64_t)k * n + r];
#pragma unroll
        for (int k = 0; k < KVOL; ++k) {
          if (p[k] >= 0) {
            float4 v = *(const float4 *)&tmp[(int64_t)p[k] * co + j];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          }
        }
      } else {
        for (int k = 0; k < kvol; ++k) {
          int32_t q = pos[(int64_t)k * n + r];
          if (q >= 0) {
            float4 v = *(const float4 *)&tmp[(int64_t)q * co + j];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          }
        }
      }
      *(float4 *)&out[r * co + j] = acc;
      s0[0] += (double)acc.x; s1[0] += (double)acc.x * (double)acc.x;
      s0[1] += (double)acc.y; s1[1] += (double)acc.y * (double)acc.y;
      s0[2] += (double)acc.z; s1[2] += (double)acc.z * (double)acc.z;
      s0[3] += (double)acc.w; s1[3] += (double)acc.w * (double)acc.w;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      sh[(0 * RL + rl) * co + j + v] = s0[v];
      sh[(1 * RL + rl) * co + j + v] = s1[v];
    }
  }
  __syncthreads();
  for (int e = tid; e < 2 * co; e += 256) {
    int which = e / co, col = e - which * co;
    double s = 0;
    for (int q = 0; q < RL; ++q) s += sh[(which * RL + q) * co + col];
    lb_store(&part[((int64_t)blockIdx.x * 2 + which) * co + col], s);
  }
  // column totals [2][co] behind the nb partial rows, by the last block to finish (ftx_lastblock.h): what ftx_bn_train_fwd_totals reads
  __syncthreads();   // sh is free again: 2 * RL * co = 2048 doubles >= 256 + 2 co for co <= 512
  last_block_totals(part, (int)gridDim.x, co, sc, sh, StoreTotals{part + (int64_t)gridDim.x * 2 * co, co});
}

extern "C" int32_t ftx_spconv_reduce_stats_blocks(int64_t n, int32_t co) {
  if (n <= 0 || co < 4) return 1;
  const int rl = 256 / (co / 4) > 0 ? 256 / (co / 4) : 1;
  int64_t b = ceil_div(n, 2 * rl);   // ~2 rows per thread
  if (b > 2048) b = 2048;
  return (int32_t)(b < 1 ? 1 : b);
}

extern "C" int ftx_spconv_reduce_stats(const float *tmp, const int32_t *pos, int64_t n, int32_t co, int32_t kvol, float *out, double *part,
                                       int32_t nb, void *stream) {
  FTX_REQUIRE(n >= 1 && kvol >= 1 && co >= 4 && co % 4 == 0 && co <= 512, "ftx_spconv_reduce_stats: bad size (co must be a multiple of 4 in [4, 512])");
  FTX_REQUIRE(pos && out && part, "ftx_spconv_reduce_stats: null pointer");
  FTX_REQUIRE(nb == ftx_spconv_reduce_stats_blocks(n, co), "ftx_spconv_reduce_stats: nb must come from ftx_spconv_reduce_stats_blocks");
  hipStream_t st = (hipStream_t)stream;
  const StreamScratch sc = stream_scratch(st);
  if (!sc.counters) return FTX_ELAUNCH;
  const int rl = 256 / (co / 4);
  const size_t lds = sizeof(double) * 2 * rl * co;
  if (kvol == 27)
    spconv_reduce_stats_kernel<27><<<nb, 256, lds, st>>>(tmp, pos, n, co, kvol, out, part, sc);
  else if (kvol == 8)
    spconv_reduce_stats_kernel<8><<<nb, 256, lds, st>>>(tmp, pos, n, co, kvol, out, part, sc);
  else
    spconv_reduce_stats_kernel<0><<<nb, 256, lds, st>>>(tmp, pos, n, co, kvol, out, part, sc);
  return check_launch("ftx_spconv_reduce_stats");
}

// ---------------------------------------------------------------------------------------
// weight gradient: dW[k] = sum_{p in k} A[idx_a[p],:]^T @ G[idx_g[p],:]
//
// Block = (tile of `tile_len` consecutive pairs of ONE offset, M tile, N tile) -> one (TM x TN) partial of dW[k]; the partials of
// an offset are summed by an ordered second pass (an offset that fits one tile is written straight into dW[k]).
//
// The reduction index is the PAIR, so the gathered rows are staged row-major ([pair][channel], 16-byte stores) and used as they
// are: a wave's (32 MI) x (32 NI) piece of the tile takes channels  base + MI*i + mi  /  base + NI*j + ni  for MFMA index i / j,
// i.e. the MI (NI) sub-tiles interleave.  One lane then needs MI (NI) CONSECUTIVE floats of a staged row per reduction step: a
// single ds_read_b64 / b128 per operand feeds MI*NI MFMAs (the previous kernel fed every MFMA from its own ds_read_b32), and the
// accumulator registers of a lane hold 4*NI consecutive g-channels of MI a-channel rows, so the epilogue is 16-byte stores.
// Waves: WMG x WNG x KS = 4; KS > 1 splits the pairs of each step and sums the KS groups through LDS in a fixed order.
// ---------------------------------------------------------------------------------------
constexpr int WG_BR = 32;       // pairs staged per step
constexpr int WG_ROUND = 1024;  // pair indices kept in LDS at a time

template <int N> struct FragLoad;
template <> struct FragLoad<1> { static __device__ __forceinline__ void ld(const float *p, float (&f)[1]) { f[0] = *p; } };
template <> struct FragLoad<2> { static __device__ __forceinline__ void ld(const float *p, float (&f)[2]) { float2 v = *(const float2 *)p; f[0] = v.x; f[1] = v.y; } };
template <> struct FragLoad<3> { static __device__ __forceinline__ void ld(const float *p, float (&f)[3]) { f[0] = p[0]; f[1] = p[1]; f[2] = p[2]; } };
template <> struct FragLoad<4> { static __device__ __forceinline__ void ld(const float *p, float (&f)[4]) { float4 v = *(const float4 *)p; f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; } };

template <int MI, int NI, int WMG, int WNG>
__global__ __launch_bounds__(256) void pairs_wgrad_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ idx_a,
                                                          const float *__restrict__ G, int64_t rows_g, const int32_t *__restrict__ idx_g,
                                                          const int32_t *__restrict__ koff, int ca, int cg, int kvol, int tile_len,
                                                          float *__restrict__ part, float *__restrict__ dW, int n_dense) {
  // idx_a == nullptr: dense mode, rows [0, n_dense) of A and G pair up one to one (kvol = 1)
  constexpr int TM = 32 * MI * WMG, TN = 32 * NI * WNG, KS = 4 / (WMG * WNG);
  constexpr int STAGE = WG_BR * (TM + TN);                 // floats
  constexpr int RED = KS > 1 ? MI * NI * 1024 * WMG * WNG : 0;  // floats: one KS group's accumulators
  constexpr int LDSF = STAGE > RED ? STAGE : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  __shared__ int32_t s_ia[WG_ROUND], s_ig[WG_ROUND];
  __shared__ uint8_t s_ok[WG_ROUND];
  __shared__ int s_tile[4];
  __shared__ int s_bad;
  float *As = lds, *Gs = lds + WG_BR * TM;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int wq = wave % (WMG * WNG), ks = wave / (WMG * WNG);
  const int wm = wq % WMG, wn = wq / WMG;
  // Tile = `tile_len` consecutive pairs of ONE offset.  Offsets differ a lot in pair count (the centre offset of a submanifold
  // map has one pair per voxel, ~7x the others), so tiles are cut from the pair list, not per offset: a wave-level scan of
  // koff maps block -> (offset, range).
  if (tid < 64) {
    const int b = blockIdx.x;
    int c = 0;
    if (tid < kvol) c = koff ? koff[tid + 1] - koff[tid] : n_dense;
    int nt = (c + tile_len - 1) / tile_len;
    int incl = nt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, 64);
      if (tid >= off) incl += v;
    }
    int excl = incl - nt;
    bool mine = (tid < kvol) && b >= excl && b < incl;
    unsigned long long msk = __ballot(mine);
    if (mine) {
      int t = b - excl;
      int first = (koff ? koff[tid] : 0) + t * tile_len;
      int left = c - t * tile_len;
      s_tile[0] = tid;
      s_tile[1] = first;
      s_tile[2] = first + (left > tile_len ? tile_len : left);
      s_tile[3] = nt;
    }
    if (msk == 0ull && tid == 0) s_tile[0] = -1;
  }
  __syncthreads();
  const int k = s_tile[0];
  if (k < 0) return;  // surplus block of the upper-bound grid
  const int m0 = blockIdx.y * TM, n0 = blockIdx.z * TN;
  const int lo = s_tile[1], hi = s_tile[2];
  const bool single = s_tile[3] == 1;   // the only tile of its offset: the result IS dW[k]

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  // Two register stages: the gathers of step s+2 are issued while step s is in the matrix cores, so a block hides its own load
  // latency (blocks of one CU start together and stay in lockstep, so relying on the other resident blocks does not work).
  //
  // The gathers are UNCONDITIONAL loads from addresses that are always valid: the staged indices are clamped to real rows and the
  // channel offset to the last float4 of a row.  What such a load brings in for a channel >= ca (cg) only ever reaches
  // accumulator rows / columns >= ca (cg), which the epilogue never stores, so padded channels need no masking at all; pairs past
  // the end of the tile (and pairs with an out-of-range index) are zeroed at LDS-store time, on a block-uniform slow path that a
  // tile enters for its last step only.  (With a branch per load, as before, every load waited for its own index read:
  // ~16 serialised LDS round trips per step.)
  constexpr int QA = TM / 32, QG = TN / 32;   // float4 per thread per step and operand
  float4 ra0[QA], rg0[QG], ra1[QA], rg1[QG];
  int rbase = lo, rend = lo;
  int pra[QA], prg[QG], ca_off[QA], cg_off[QG];
#pragma unroll
  for (int q = 0; q < QA; ++q) {
    int e = q * 256 + tid;
    pra[q] = e / (TM / 4);
    int c = m0 + (e - pra[q] * (TM / 4)) * 4;
    ca_off[q] = c + 4 <= ca ? c : ca - 4;
  }
#pragma unroll
  for (int q = 0; q < QG; ++q) {
    int e = q * 256 + tid;
    prg[q] = e / (TN / 4);
    int c = n0 + (e - prg[q] * (TN / 4)) * 4;
    cg_off[q] = c + 4 <= cg ? c : cg - 4;
  }
  auto load_step = [&](int p0, float4 (&ra)[QA], float4 (&rg)[QG]) {
    const int o = p0 - rbase;
    if (o >= WG_ROUND) return;   // past this round's indices: the step is never consumed
    int32_t ia[QA], ig[QG];
#pragma unroll
    for (int q = 0; q < QA; ++q) ia[q] = s_ia[(o + pra[q]) & (WG_ROUND - 1)];
#pragma unroll
    for (int q = 0; q < QG; ++q) ig[q] = s_ig[(o + prg[q]) & (WG_ROUND - 1)];
#pragma unroll
    for (int q = 0; q < QA; ++q) ra[q] = *(const float4 *)(A + ((int64_t)ia[q] * ca + ca_off[q]));
#pragma unroll
    for (int q = 0; q < QG; ++q) rg[q] = *(const float4 *)(G + ((int64_t)ig[q] * cg + cg_off[q]));
  };
  auto store_step = [&](int p0, float4 (&ra)[QA], float4 (&rg)[QG]) {
    if (p0 + WG_BR > rend || s_bad) {   // block-uniform: last step of the tile (or a malformed pair list)
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        const int p = p0 + pra[q];
        if (p >= rend || s_ok[(p - rbase) & (WG_ROUND - 1)] == 0) ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < QG; ++q) {
        const int p = p0 + prg[q];
        if (p >= rend || s_ok[(p - rbase) & (WG_ROUND - 1)] == 0) rg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int q = 0; q < QA; ++q) *(float4 *)&As[(q * 256 + tid) * 4] = ra[q];   // row-major [pair][TM]: e*4 == pr*TM + c4
#pragma unroll
    for (int q = 0; q < QG; ++q) *(float4 *)&Gs[(q * 256 + tid) * 4] = rg[q];
  };
  const float *ap = As + wm * 32 * MI + MI * l31, *gp = Gs + wn * 32 * NI + NI * l31;
  auto mfma_step = [&]() {
    constexpr int ITS = WG_BR / 2 / KS;
    float af[2][MI], gf[2][NI];
    auto frag = [&](int buf, int it) {
      const int kk = 2 * (it * KS + ks) + half;
      FragLoad<MI>::ld(ap + kk * TM, af[buf]);
      FragLoad<NI>::ld(gp + kk * TN, gf[buf]);
    };
    frag(0, 0);
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      if (it + 1 < ITS) frag((it + 1) & 1, it + 1);   // next fragments in flight under this step's MFMAs
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(gf[it & 1][j], af[it & 1][i], acc[i][j], 0, 0, 0);  // rows: g-channel, cols: a-channel
    }
  };

  for (rbase = lo; rbase < hi; rbase += WG_ROUND) {
    rend = (rbase + WG_ROUND < hi) ? rbase + WG_ROUND : hi;
    __syncthreads();  // previous round's gathers are done with s_ia / s_ig
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int t = tid; t < WG_ROUND; t += 256) {
      // every slot gets a loadable row: slots past the end repeat row 0, malformed pairs are flagged and zeroed at store time
      int32_t ia = 0, ig = 0;
      uint8_t ok = 0;
      if (t < rend - rbase) {
        ia = idx_a ? idx_a[rbase + t] : rbase + t;
        ig = idx_g ? idx_g[rbase + t] : rbase + t;
        ok = 1;
        if (ia < 0 || ia >= rows_a || ig < 0 || ig >= rows_g) {
          ia = ig = 0;
          ok = 0;
          s_bad = 1;
        }
      }
      s_ia[t] = ia;
      s_ig[t] = ig;
      s_ok[t] = ok;
    }
    __syncthreads();
    load_step(rbase, ra0, rg0);
    load_step(rbase + WG_BR, ra1, rg1);
    for (int p0 = rbase; p0 < rend; p0 += 2 * WG_BR) {
      store_step(p0, ra0, rg0);
      __syncthreads();
      load_step(p0 + 2 * WG_BR, ra0, rg0);
      mfma_step();
      __syncthreads();
      if (p0 + WG_BR >= rend) break;
      store_step(p0 + WG_BR, ra1, rg1);
      __syncthreads();
      load_step(p0 + 3 * WG_BR, ra1, rg1);
      mfma_step();
      __syncthreads();
    }
  }

  if (KS > 1) {  // sum the pair-subsets of the KS wave groups, fixed order (the staging buffers are free now)
    float *red = lds;
    for (int r = 1; r < KS; ++r) {
      if (ks == r) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) red[(((wq * MI + i) * NI + j) * 16 + g) * 64 + lane] = acc[i][j][g];
      }
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] += red[(((wq * MI + i) * NI + j) * 16 + g) * 64 + lane];
      }
      __syncthreads();
    }
  }

  if (ks == 0) {
    const int64_t mat = (int64_t)ca * cg;
    float *dst = single ? dW + (int64_t)k * mat : part + (int64_t)blockIdx.x * mat;   // tiles are numbered in offset order
    // accumulator register 4q + e of lane (l31, half) is g-row 8q + 4*half + e of sub-tile (i, j): g-channel gb + NI*(8q+4half+e) + j,
    // a-channel ab + MI*l31 + i.  For fixed (i, q) the NI*4 values over (e, j) are consecutive g-channels: 16-byte stores.
    const int ab = m0 + wm * 32 * MI + MI * l31, gb = n0 + wn * 32 * NI;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = ab + i;
      if (row < ca) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4 * NI];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NI; ++j) v[e * NI + j] = acc[i][j][4 * q + e];
          const int col = gb + NI * (8 * q + 4 * half);
#pragma unroll
          for (int t = 0; t < NI; ++t)
            if (col + 4 * t < cg) *(float4 *)&dst[(int64_t)row * cg + col + 4 * t] = make_float4(v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]);
        }
      }
    }
  }
}

// dW[k] = sum of the partial tiles of offset k (tiles are numbered in offset order); an offset with ONE tile was written by the
// main kernel itself.  Block = (256/TL) float4 columns x TL tile lanes; lane l sums tiles l, l+TL, ... and the TL lane sums are
// added in lane order through LDS: a fixed summation tree, bit-reproducible.
template <int TL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, const int32_t *__restrict__ koff, int kvol,
                                                           int tile_len, int n_dense, int64_t mat, float *__restrict__ dW) {
  constexpr int COLS = 256 / TL;
  __shared__ float4 sh[TL][COLS];
  const int k = blockIdx.y;
  int first = 0, cnt = 0;
  for (int q = 0; q <= k; ++q) {
    int c = koff ? koff[q + 1] - koff[q] : n_dense;
    int nt = (c + tile_len - 1) / tile_len;
    if (q < k) first += nt; else cnt = nt;
  }
  if (cnt == 1) return;   // written directly by pairs_wgrad_kernel (block-uniform exit)
  const int col = threadIdx.x % COLS, tl = threadIdx.x / COLS;
  const int64_t chunks = ceil_div(mat / 4, COLS);
  // a block walks several column chunks: thousands of 4-KB blocks are bound by workgroup dispatch, not by bytes
  for (int64_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    const int64_t e = (chunk * COLS + col) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < mat) {
      const float *src = part + (int64_t)first * mat + e;
      int t = tl;
      for (; t + 3 * TL < cnt; t += 4 * TL) {   // four independent loads in flight, added in tile order
        float4 v0 = *(const float4 *)&src[(int64_t)t * mat], v1 = *(const float4 *)&src[(int64_t)(t + TL) * mat];
        float4 v2 = *(const float4 *)&src[(int64_t)(t + 2 * TL) * mat], v3 = *(const float4 *)&src[(int64_t)(t + 3 * TL) * mat];
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
        s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
        s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
      }
      for (; t < cnt; t += TL) {
        float4 v = *(const float4 *)&src[(int64_t)t * mat];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    if (TL > 1) {
      sh[tl][col] = s;
      __syncthreads();
      if (tl == 0) {
#pragma unroll
        for (int l = 1; l < TL; ++l) {
          float4 v = sh[l][col];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
      }
      __syncthreads();
    }
    if (tl == 0 && e < mat) *(float4 *)&dW[(int64_t)k * mat + e] = s;
  }
}

// Tile shape per channel count.  M side: 32 / 64 / 96 (multiples of 96 that are not multiples of 128: 96, 192) / 128;
// N side: 32 / 64 / 96 / 128.
struct WgradCfg { int mi, wmg, ni, wng; };
static WgradCfg wgrad_config(int ca, int cg) {
  WgradCfg c;
  if (ca <= 32) { c.mi = 1; c.wmg = 1; }
  else if (ca <= 64) { c.mi = 2; c.wmg = 1; }
  else if (ca % 96 == 0 && ca % 128 != 0) { c.mi = 3; c.wmg = 1; }
  else { c.mi = 2; c.wmg = 2; }
  if (cg <= 32) { c.ni = 1; c.wng = 1; }
  else if (cg <= 64) { c.ni = 2; c.wng = 1; }
  else if (cg % 96 == 0 && cg % 128 != 0) { c.ni = 3; c.wng = 1; }
  else { c.ni = 2; c.wng = 2; }
  // a 96 x 96 tile per wave is 9 accumulators (144 registers): one wave per SIMD.  96 -> 96 layers take a 128 x 96 tile instead
  // (2 x 3 accumulators per wave, the last 32 M rows are padding).
  if (c.mi == 3 && c.ni == 3) { c.mi = 2; c.wmg = 2; }
  return c;
}

// Resident blocks per CU of the instantiation a layer uses, as a TABLE: registers decide (a block is one wave per SIMD, a SIMD has 512
// vector registers; LDS never binds first).  Values = min(8, 512 / allocated VGPRs) read from the gfx950 code object (llvm-readelf --notes:
// .vgpr_count 60 / 96-104 / 128-136 / 156 / 184-236 / 316) -- tests/test_cabi.py recomputes them from the built object and fails when the
// table is stale.  A table and not hipOccupancyMaxActiveBlocksPerMultiprocessor: the tile length, the workspace size and the summation tree
// of the weight gradient (hence its bits) must be functions of the arguments alone, the same on every host and device (round 2 asked the
// runtime, with a fallback of 2 where there was no device to ask).
static int wgrad_occ(const WgradCfg &c) {
  // rows: M side (mi, wmg) = (1,1) (2,1) (3,1) (2,2); columns: N side (ni, wng) in the same order
  static const int occ[4][4] = {{8, 4, 3, 3}, {5, 3, 2, 2}, {4, 2, 1, 2}, {3, 2, 2, 2}};
  auto side = [](int i, int w) { return w == 2 ? 3 : i - 1; };
  return occ[side(c.mi, c.wmg)][side(c.ni, c.wng)];
}
// (mi, wmg, ni, wng) -> table value, for the build-time check of the table against the code object
extern "C" int32_t ftx_spconv_wgrad_table_blocks(int32_t mi, int32_t wmg, int32_t ni, int32_t wng) {
  if (!((mi >= 1 && mi <= 3 && wmg == 1) || (mi == 2 && wmg == 2)) || !((ni >= 1 && ni <= 3 && wng == 1) || (ni == 2 && wng == 2))) return -1;
  WgradCfg c{mi, wmg, ni, wng};
  return wgrad_occ(c);
}
extern "C" int32_t ftx_spconv_wgrad_resident_blocks(int32_t ca, int32_t cg) { return wgrad_occ(wgrad_config(ca, cg)); }
constexpr int WGRAD_CUS = 256;   // MI355X; a constant of the tiling, not a device query (see above)

// Pairs per tile.  All blocks of a launch should be resident together: a launch of 1.2x the resident slots takes as long as one of 2x
// (measured: 620 blocks on 512 slots ran 1.7x longer than 820).  So the tile length is chosen for R full rounds of
// slots = CUs x resident blocks per CU, R as small as keeps a tile <= 4096 pairs; every offset adds about half a tile of rounding.
static int wgrad_tile_len(int64_t n_pairs, int ca, int cg, int kvol) {
  const WgradCfg c = wgrad_config(ca, cg);
  const int64_t mn_tiles = ceil_div(ca, 32 * c.mi * c.wmg) * ceil_div(cg, 32 * c.ni * c.wng);
  const int64_t slots = (int64_t)WGRAD_CUS * wgrad_occ(c);
  int64_t len = 256;
  for (int rounds = 1; rounds <= 64; ++rounds) {
    int64_t tiles = (slots * rounds * 15 / 16) / mn_tiles - (kvol + 1) / 2;   // 1/16 of head room: an overshoot costs a whole round
    if (tiles < 1) tiles = 1;
    len = ceil_div(ceil_div(n_pairs, tiles), 2 * WG_BR) * 2 * WG_BR;
    if (len <= 4096) break;
  }
  if (len < 256) len = 256;
  return (int)len;
}

static int64_t wgrad_tiles_ub(int64_t n_pairs, int tile_len, int kvol) { return ceil_div(n_pairs, tile_len) + kvol; }

extern "C" size_t ftx_spconv_pairs_wgrad_workspace_bytes(int64_t n_pairs, int32_t ca, int32_t cg, int32_t kvol) {
  if (n_pairs <= 0 || ca <= 0 || cg <= 0 || kvol <= 0) return 256;
  int len = wgrad_tile_len(n_pairs, ca, cg, kvol);
  return sizeof(float) * (size_t)wgrad_tiles_ub(n_pairs, len, kvol) * ca * cg;
}

template <int MI, int WMG>
static void launch_wgrad_n(const WgradCfg &c, dim3 grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *idx_a, const float *G, int64_t rows_g,
                           const int32_t *idx_g, const int32_t *koff, int ca, int cg, int kvol, int tl, float *part, float *dW, int n_dense) {
  if (c.ni == 1)
    pairs_wgrad_kernel<MI, 1, WMG, 1><<<grid, 256, 0, st>>>(A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tl, part, dW, n_dense);
  else if (c.ni == 3)
    pairs_wgrad_kernel<MI, 3, WMG, 1><<<grid, 256, 0, st>>>(A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tl, part, dW, n_dense);
  else if (c.wng == 1)
    pairs_wgrad_kernel<MI, 2, WMG, 1><<<grid, 256, 0, st>>>(A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tl, part, dW, n_dense);
  else
    pairs_wgrad_kernel<MI, 2, WMG, 2><<<grid, 256, 0, st>>>(A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tl, part, dW, n_dense);
}

extern "C" int ftx_spconv_pairs_wgrad(const float *A, int64_t rows_a, const int32_t *idx_a, const float *G, int64_t rows_g, const int32_t *idx_g,
                                      const int32_t *koff, int64_t n_pairs, int32_t ca, int32_t cg, int32_t kvol, float *dW, void *workspace,
                                      size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n_pairs >= 0 && rows_a >= 0 && rows_g >= 0 && kvol >= 1 && kvol <= 64, "ftx_spconv_pairs_wgrad: bad size");
  FTX_REQUIRE(ca >= 4 && ca % 4 == 0 && cg >= 4 && cg % 4 == 0, "ftx_spconv_pairs_wgrad: channels must be multiples of 4 (ca=%d cg=%d)", ca, cg);
  FTX_REQUIRE(dW, "ftx_spconv_pairs_wgrad: null dW");
  hipStream_t st = (hipStream_t)stream;
  const int64_t mat = (int64_t)ca * cg;
  if (n_pairs == 0) {
    if (hipMemsetAsync(dW, 0, sizeof(float) * kvol * mat, st) != hipSuccess) return check_launch("ftx_spconv_pairs_wgrad memset");
    return FTX_OK;
  }
  FTX_REQUIRE(A && G && rows_a >= 1 && rows_g >= 1, "ftx_spconv_pairs_wgrad: null pointer or empty operand");
  const bool dense = (idx_a == nullptr && idx_g == nullptr && koff == nullptr);
  FTX_REQUIRE(dense || (idx_a && idx_g && koff), "ftx_spconv_pairs_wgrad: idx_a, idx_g and koff must be all set or all null (dense rows)");
  FTX_REQUIRE(!dense || (kvol == 1 && n_pairs <= rows_a && n_pairs <= rows_g), "ftx_spconv_pairs_wgrad: dense mode needs kvol == 1 and n_pairs rows in A and G");
  FTX_REQUIRE(n_pairs < 0x7fffffff, "ftx_spconv_pairs_wgrad: too many pairs");
  const int tile_len = wgrad_tile_len(n_pairs, ca, cg, kvol);
  const int64_t tiles = wgrad_tiles_ub(n_pairs, tile_len, kvol);
  size_t need = sizeof(float) * (size_t)tiles * mat;
  if (!workspace || workspace_bytes < need) {
    set_error("ftx_spconv_pairs_wgrad: workspace %zu < required %zu", workspace_bytes, need);
    return FTX_EWORKSPACE;
  }
  float *part = (float *)workspace;
  const WgradCfg c = wgrad_config(ca, cg);
  dim3 grid((unsigned)tiles, (unsigned)ceil_div(ca, 32 * c.mi * c.wmg), (unsigned)ceil_div(cg, 32 * c.ni * c.wng));
  if (c.mi == 1)
    launch_wgrad_n<1, 1>(c, grid, st, A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tile_len, part, dW, (int)n_pairs);
  else if (c.mi == 3)
    launch_wgrad_n<3, 1>(c, grid, st, A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tile_len, part, dW, (int)n_pairs);
  else if (c.wmg == 1)
    launch_wgrad_n<2, 1>(c, grid, st, A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tile_len, part, dW, (int)n_pairs);
  else
    launch_wgrad_n<2, 2>(c, grid, st, A, rows_a, idx_a, G, rows_g, idx_g, koff, ca, cg, kvol, tile_len, part, dW, (int)n_pairs);
  // the centre offset of a submanifold map holds ~6x the average pair count: size the tile lanes for it, not for the average
  const int64_t big_tiles = kvol > 1 ? 6 * tiles / kvol : tiles;
  const int64_t want_blocks = ceil_div(1024, kvol);   // ~4 blocks per CU over all offsets
  auto rgrid = [&](int cols) { int64_t c = ceil_div(mat / 4, cols); return dim3((unsigned)(c < want_blocks ? c : want_blocks), (unsigned)kvol); };
  if (big_tiles <= 4)
    wgrad_reduce_kernel<1><<<rgrid(256), 256, 0, st>>>(part, koff, kvol, tile_len, (int)n_pairs, mat, dW);
  else if (big_tiles <= 32)
    wgrad_reduce_kernel<4><<<rgrid(64), 256, 0, st>>>(part, koff, kvol, tile_len, (int)n_pairs, mat, dW);
  else
    wgrad_reduce_kernel<16><<<rgrid(16), 256, 0, st>>>(part, koff, kvol, tile_len, (int)n_pairs, mat, dW);
  return check_launch("ftx_spconv_pairs_wgrad");
}
