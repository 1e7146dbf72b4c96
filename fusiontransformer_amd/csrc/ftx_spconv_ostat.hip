// Output-stationary sparse convolution for the thin layers (c_out <= 64) of SPVCNN (models/spvcnn.py:22-35,98-126): ONE launch,
// no pair-row scratch (`tmp`), no reduce pass, the BatchNorm statistics of the result from the same blocks.
//
//   out[o,:] = sum over k (ascending) of A[nbr[k,o],:] @ W[k]          nbr (K, N_out): input row of output row o at offset k, or -1
//
// The pair-list form (ftx_spconv.hip) writes one row of `tmp` per (k, o) pair and reads it back in the reduce pass; for a layer with
// 32 / 64 channels those two streams are 60 % of its traffic and the layer is HBM-, not matrix-bound (8-16 flop / byte).  Here a WAVE
// owns 64 consecutive output rows and keeps their accumulators in LDS:
//
//   per offset k:  the 64 lanes read nbr[k, r0 + lane] (one coalesced 256-byte load, prefetched one offset ahead), the valid ones are
//                  compacted by ballot + prefix count into a list of (input row, local output row) -- ~11 of 64 for an off-centre
//                  offset, all 64 for the centre --, and every 16 entries become one 16 x c_out tile on the matrix cores
//                  (v_mfma_f32_16x16x4_f32: W[k] is the row operand, the gathered rows the column operand, both loaded straight from
//                  global memory / L2 into registers: lane (pair p, group g) needs channels 8t + 4(g&1) + (g>>1) (+2) of its pair's row,
//                  i.e. elements of ONE 16-byte load per 8 channels), and the tile is added into the LDS rows of its pairs.
//   at the end:    the 64 x c_out block is written once, and its column sums / sums of squares (float64, fixed order) go to the
//                  last-block hand-over of ftx_lastblock.h exactly as spconv_reduce_stats_kernel's do.
//
// A 16-wide tile holds 11 pairs on average instead of the 5.6-of-32 an output-stationary 32-row tile would (the voxel rows are in hash
// order, i.e. spatially random: a voxel has 5-9 of 27 neighbours), and nothing is padded in memory.  Waves never synchronise with each
// other before the statistics, so a CU keeps 8-20 independent gather streams in flight.
//
// Bit-identical to pairs_gemm + reduce: an f32 MFMA is a k-ordered fmaf chain (cdna_hip_programming.md), the four k slots of a
// 16x16x4 instruction are given the channels the pair kernel's 32x32x2 sequence consumes in the same order -- (8t, 8t+4, 8t+1, 8t+5),
// then (8t+2, 8t+6, 8t+3, 8t+7) --, and an output row receives its offsets in ascending k (tests/test_ops_gpu.py checks torch.equal).
//
// `flip`: data gradient of a submanifold (stride-1, odd kernel) convolution.  Its map is symmetric -- nbr[k, o] = i  <=>
// nbr[K-1-k, i] = o -- so gin[i] = sum_k gout[nbr[K-1-k, i]] @ W[k]^T is the same kernel on the same table, read in mirrored order.
#include "ftx_common.h"
#include "ftx_lastblock.h"

using namespace ftx;

typedef float os_f32x4 __attribute__((ext_vector_type(4)));

constexpr int OS_ROWS = 64;   // output rows per wave
constexpr int OS_WAVES = 4;   // independent waves per block (they only meet for the statistics)

template <int CA, int CO16, bool WT>
__global__ __launch_bounds__(256) void spconv_ostat_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ nbr, int64_t n_out,
                                                           const float *__restrict__ W, int flip, int kvol, float *__restrict__ out,
                                                           double *part, StreamScratch sc) {
  constexpr int CO = 16 * CO16, ST = CO + 4;   // LDS row stride in floats: 36 / 68 spreads 16 random rows over the banks
  constexpr int NCH = (CA + 31) / 32;          // 32-channel chunks of the reduction
  extern __shared__ __attribute__((aligned(16))) float os_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p = lane & 15, g = lane >> 4;
  float *acc = os_smem + wave * OS_ROWS * ST;
  int32_t *list = (int32_t *)(os_smem + OS_WAVES * OS_ROWS * ST) + wave * 64;
  const int64_t r0 = ((int64_t)blockIdx.x * OS_WAVES + wave) * OS_ROWS;

  for (int e = lane; e < OS_ROWS * ST / 4; e += 64) ((float4 *)acc)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  __builtin_amdgcn_wave_barrier();

  if (r0 < n_out) {   // wave-uniform
    const bool row_ok = r0 + lane < n_out;
    const int64_t my = row_ok ? r0 + lane : n_out - 1;
    auto nbr_at = [&](int k) { return nbr[(int64_t)(flip ? kvol - 1 - k : k) * n_out + my]; };
    int32_t v_next = nbr_at(0);
    for (int k = 0; k < kvol; ++k) {
      int32_t v = v_next;
      if (k + 1 < kvol) v_next = nbr_at(k + 1);
      const bool valid = row_ok && v >= 0 && v < rows_a;
      const unsigned long long mask = __ballot(valid);
      if (mask == 0ull) continue;
      const int cnt = __popcll(mask);
      if (valid) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        list[rank] = (v << 6) | lane;
      }
      __builtin_amdgcn_wave_barrier();
      const float *Wk = W + (int64_t)k * CA * CO;
      for (int t0 = 0; t0 < cnt; t0 += 16) {
        const bool ok = t0 + p < cnt;
        const int32_t ent = list[ok ? t0 + p : 0];   // entry 0 exists (cnt >= 1): a padded lane gathers a real row and is never stored
        const float *arow = A + (int64_t)(ent >> 6) * CA;
        os_f32x4 c[CO16];
#pragma unroll
        for (int j = 0; j < CO16; ++j) c[j] = (os_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          constexpr int TS = (CA >= 32) ? 4 : (CA + 7) / 8;   // 8-channel steps in this chunk
          float4 a[TS];
#pragma unroll
          for (int t = 0; t < TS; ++t) {
            const int col = ch * 32 + 8 * t + 4 * (g & 1);
            a[t] = (col + 4 <= CA) ? *(const float4 *)(arow + col) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int t = 0; t < TS; ++t) {
            const int cb = ch * 32 + 8 * t + 4 * (g & 1);   // this lane group's 4-channel base
            const float a1 = (g >> 1) ? a[t].y : a[t].x;    // channel cb + (g>>1)
            const float a2 = (g >> 1) ? a[t].w : a[t].z;    // channel cb + (g>>1) + 2
#pragma unroll
            for (int j = 0; j < CO16; ++j) {
              float w1 = 0.f, w2 = 0.f;
              if (cb + 4 <= CA) {
                if (WT) {   // W[k] stored (CO, CA): four consecutive reduction channels in one 16-byte load
                  const float4 wv = *(const float4 *)(Wk + (int64_t)(16 * j + p) * CA + cb);
                  w1 = (g >> 1) ? wv.y : wv.x;
                  w2 = (g >> 1) ? wv.w : wv.z;
                } else {    // W[k] stored (CA, CO): 16 lanes read 64 contiguous bytes of a row
                  w1 = Wk[(int64_t)(cb + (g >> 1)) * CO + 16 * j + p];
                  w2 = Wk[(int64_t)(cb + (g >> 1) + 2) * CO + 16 * j + p];
                }
              }
              c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, a1, c[j], 0, 0, 0);
              c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2, a2, c[j], 0, 0, 0);
            }
          }
        }
        if (ok) {   // lane (pair p, group g) holds channels 16j + 4g .. + 3 of its pair: one 16-byte read-modify-write per column block
          float *dst = acc + (ent & 63) * ST + 4 * g;
#pragma unroll
          for (int j = 0; j < CO16; ++j) {
            float4 d = *(float4 *)(dst + 16 * j);
            d.x += c[j][0]; d.y += c[j][1]; d.z += c[j][2]; d.w += c[j][3];
            *(float4 *)(dst + 16 * j) = d;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // the wave's 64 x CO block, written once
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < OS_ROWS * (CO / 4); e += 64) {
      const int row = e / (CO / 4), c4 = e - row * (CO / 4);
      if (r0 + row < n_out) *(float4 *)(out + (r0 + row) * CO + 4 * c4) = *(const float4 *)(acc + row * ST + 4 * c4);
    }
  }
  if (part == nullptr) return;   // launch-uniform

  // BatchNorm statistics of the block's 256 rows: column sums and sums of squares in float64, rows in order within a wave, waves in
  // order within the block, blocks in order by the last block to finish (ftx_lastblock.h).  Rows past n_out are zero rows.
  __shared__ double os_part[OS_WAVES][2][64];
  if (lane < CO) {
    double s0 = 0, s1 = 0;
    for (int r = 0; r < OS_ROWS; ++r) {
      const double x = (double)acc[r * ST + lane];
      s0 += x;
      s1 += x * x;
    }
    os_part[wave][0][lane] = s0;
    os_part[wave][1][lane] = s1;
  }
  __syncthreads();
  for (int e = tid; e < 2 * CO; e += 256) {
    const int which = e / CO, col = e - which * CO;
    double s = os_part[0][which][col];
#pragma unroll
    for (int w = 1; w < OS_WAVES; ++w) s += os_part[w][which][col];
    lb_store(&part[((int64_t)blockIdx.x * 2 + which) * CO + col], s);
  }
  __syncthreads();   // the accumulators are dead: their LDS is the hand-over's scratch (256 + 2 CO doubles <= 3 KB)
  last_block_totals(part, (int)gridDim.x, CO, sc, (double *)os_smem, StoreTotals{part + (int64_t)gridDim.x * 2 * CO, CO});
}

namespace {
constexpr size_t os_lds_bytes(int co) { return sizeof(float) * (size_t)OS_WAVES * OS_ROWS * (co + 4) + sizeof(int32_t) * OS_WAVES * 64; }

template <int CA, int CO16, bool WT>
int os_launch(unsigned grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *nbr, int64_t n_out, const float *W, int flip, int kvol,
              float *out, double *part, StreamScratch sc) {
  constexpr size_t lds = os_lds_bytes(16 * CO16);
  if (lds > 64 * 1024) {
    static bool configured = false;   // idempotent: a race sets the same attribute twice
    if (!configured) {
      if (hipFuncSetAttribute((const void *)spconv_ostat_kernel<CA, CO16, WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        set_error("ftx_spconv_ostat: cannot raise the LDS limit to %zu bytes (%s)", lds, hipGetErrorString(hipGetLastError()));
        return FTX_ELAUNCH;
      }
      configured = true;
    }
  }
  spconv_ostat_kernel<CA, CO16, WT><<<grid, 256, lds, st>>>(A, rows_a, nbr, n_out, W, flip, kvol, out, part, sc);
  return FTX_OK;
}
}  // namespace

// Layers this kernel takes: 4 / 32 / 64 input channels (4: the stem, plain layout only), 32 / 64 output channels, any kernel volume <= 64.
extern "C" int32_t ftx_spconv_ostat_supported(int32_t ca, int32_t co, int32_t kvol, int32_t w_transposed) {
  if (kvol < 1 || kvol > 64) return 0;
  if (co != 32 && co != 64) return 0;
  if (ca == 4) return (!w_transposed && co == 32) ? 1 : 0;
  return (ca == 32 || ca == 64) ? 1 : 0;
}

// Blocks of a launch over n_out rows = partial rows of the statistics (`part` holds nb rows of [2][co] doubles plus one totals row).
extern "C" int32_t ftx_spconv_ostat_blocks(int64_t n_out) {
  const int64_t b = ceil_div(n_out < 1 ? 1 : n_out, (int64_t)OS_ROWS * OS_WAVES);
  return (int32_t)b;
}

extern "C" int ftx_spconv_ostat(const float *A, int64_t rows_a, const int32_t *nbr, int64_t n_out, const float *W, int32_t w_transposed, int32_t flip,
                                int32_t ca, int32_t co, int32_t kvol, float *out, double *part, int32_t nb, void *stream) {
  FTX_REQUIRE(ftx_spconv_ostat_supported(ca, co, kvol, w_transposed), "ftx_spconv_ostat: unsupported layer (ca=%d co=%d kvol=%d transposed=%d)", ca, co, kvol, w_transposed);
  FTX_REQUIRE(n_out >= 0 && rows_a >= 0, "ftx_spconv_ostat: bad size");
  if (n_out == 0) return FTX_OK;
  FTX_REQUIRE(A && nbr && W && out && rows_a >= 1, "ftx_spconv_ostat: null pointer or empty operand");
  FTX_REQUIRE(rows_a < (1ll << 25), "ftx_spconv_ostat: more than 2^25 input rows");
  FTX_REQUIRE(!flip || kvol % 2 == 1, "ftx_spconv_ostat: flip needs an odd kernel volume (a symmetric submanifold map)");
  const int64_t blocks = ftx_spconv_ostat_blocks(n_out);
  FTX_REQUIRE(blocks <= LB_GROUP * LB_MAX_GROUPS, "ftx_spconv_ostat: too many rows for one launch");
  FTX_REQUIRE(part == nullptr || nb == (int32_t)blocks, "ftx_spconv_ostat: nb must come from ftx_spconv_ostat_blocks");
  hipStream_t st = (hipStream_t)stream;
  StreamScratch sc{nullptr, nullptr};
  if (part) {
    sc = stream_scratch(st);
    if (!sc.counters) return FTX_ELAUNCH;
  }
  const unsigned grid = (unsigned)blocks;
  int rc = FTX_OK;
#define OS_CASE(CA_, CO16_, WT_) rc = os_launch<CA_, CO16_, WT_>(grid, st, A, rows_a, nbr, n_out, W, flip, kvol, out, part, sc)
  const int co16 = co / 16;
  if (ca == 4) OS_CASE(4, 2, false);
  else if (ca == 32 && co16 == 2 && !w_transposed) OS_CASE(32, 2, false);
  else if (ca == 32 && co16 == 2) OS_CASE(32, 2, true);
  else if (ca == 32 && !w_transposed) OS_CASE(32, 4, false);
  else if (ca == 32) OS_CASE(32, 4, true);
  else if (co16 == 2 && !w_transposed) OS_CASE(64, 2, false);
  else if (co16 == 2) OS_CASE(64, 2, true);
  else if (!w_transposed) OS_CASE(64, 4, false);
  else OS_CASE(64, 4, true);
#undef OS_CASE
  if (rc != FTX_OK) return rc;
  return check_launch("ftx_spconv_ostat");
}
