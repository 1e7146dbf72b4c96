// Output-stationary sparse convolution for thin layers (c_out <= 64) of SPVCNN (models/spvcnn.py:22-35,98-126): ONE launch,
// no pair-row scratch (`tmp`), no reduce pass, the BatchNorm statistics of the result from the same blocks.
//
//   out[o,:] = sum over k (ascending) of A[nbr[k,o],:] @ W[k]          nbr (K, N_out): input row of output row o at offset k, or -1
//
// The pair-list form (ftx_spconv.hip) writes one row of `tmp` per (k, o) pair and reads it back in the reduce pass; for a layer with
// 32 channels those two streams are 60 % of its traffic and the layer is HBM-, not matrix-bound (8 flop / byte).  Here a BLOCK owns
// 64 consecutive output rows and keeps their accumulators in LDS; its waves take different offsets k at the same time:
//
//   per offset k:  the 64 lanes of a wave read nbr[k, r0 + lane] (one coalesced 256-byte load, prefetched one offset ahead), the valid
//                  ones are compacted by ballot + prefix count into a list of (input row, local output row) -- ~11 of 64 for an
//                  off-centre offset, all 64 for the centre --, and every 16 entries become one 16 x c_out tile on the matrix cores
//                  (v_mfma_f32_16x16x4_f32: W[k] is the row operand, the gathered rows the column operand, both loaded straight from
//                  global memory / L2 into registers: lane (pair p, group g) needs channels 8t + 4(g&1) + (g>>1) (+2) of its pair's row,
//                  i.e. elements of ONE 16-byte load per 8 channels); the tile is added into the LDS rows of its pairs when it is the
//                  offset's turn (see the kernel).
//   at the end:    the 64 x c_out block is written once, and its column sums / sums of squares (float64, fixed order) go to the
//                  last-block hand-over of ftx_lastblock.h exactly as spconv_reduce_stats_kernel's do.
//
// A 16-wide tile holds 11 pairs on average instead of the 5.6-of-32 an output-stationary 32-row tile would (the voxel rows are in hash
// order, i.e. spatially random: a voxel has 5-9 of 27 neighbours), and nothing is padded in memory.
//
// Measured on MI355X (tools/bench_spconv.py, batch-4 workload, us; pair-list GEMM + reduce in brackets): 4->32 at 81 k rows 18 (42),
// 32->32 at 81 k / 43 k rows 31 (44) / 25 (27), strided 2^3 32->32 10 (12) / 9 (10) -- and it LOSES where a tile needs more than
// ~250 instructions of bookkeeping per ~11 pairs: 32->64 29 (26), 64->64 56 (32), every transposed-W (data-gradient) form, 51 (44) at 32->32.
// The kernel is bound by instruction issue, not by memory or the matrix pipe: with the gathers, the W loads, the MFMAs and the turn all
// switched off it still takes 21 us at 81 k rows (~23 issued instructions per pair).  So the host uses it for c_in <= 32, c_out = 32
// forward convolutions only (functional.ostat_preferred); the entry point accepts the wider set and stays bit-identical on all of it.
//
// Bit-identical to pairs_gemm + reduce: an f32 MFMA is a k-ordered fmaf chain (cdna_hip_programming.md), the four k slots of a
// 16x16x4 instruction are given the channels the pair kernel's 32x32x2 sequence consumes in the same order -- (8t, 8t+4, 8t+1, 8t+5),
// then (8t+2, 8t+6, 8t+3, 8t+7) --, and an output row receives its offsets in ascending k (tests/test_ops_gpu.py checks torch.equal).
//
// `flip`: data gradient of a submanifold (stride-1, odd kernel) convolution.  Its map is symmetric -- nbr[k, o] = i  <=>
// nbr[K-1-k, i] = o -- so gin[i] = sum_k gout[nbr[K-1-k, i]] @ W[k]^T is the same kernel on the same table, read in mirrored order.
#include <cstdlib>
#include "ftx_common.h"
#include "ftx_lastblock.h"

using namespace ftx;

typedef float os_f32x4 __attribute__((ext_vector_type(4)));

constexpr int OS_ROWS = 64;   // output rows per block

// One block = 64 consecutive output rows, accumulators [64][CO] in LDS.  The four waves work on DIFFERENT offsets at the same time --
// neighbour indices, compaction, gathers and MFMAs of up to four offsets in flight per block, 20-32 per CU -- and only the short
// read-modify-write of the accumulator rows is ordered: a wave adds the tiles of offset k when `turn` == k and then passes the turn on,
// so every output row still receives its offsets in ascending k (the bits of the pair-list path).  A wave that finds no pair for its
// offset passes the turn without touching the accumulators.  All waves of a block are resident together, the wave holding the turn
// never waits for anything but its own loads, so the hand-over cannot deadlock; the wait is bounded anyway.
template <int CA, int CO16, bool WT, int OS_WAVES>
__global__ __launch_bounds__(64 * OS_WAVES) void spconv_ostat_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ nbr, int64_t n_out,
                                                           const float *__restrict__ W, int flip, int kvol, float *__restrict__ out,
                                                           double *part, StreamScratch sc) {
  constexpr int CO = 16 * CO16, ST = CO + 4;   // LDS row stride in floats: 36 / 68 spreads 16 random rows over the banks
  constexpr int NCH = (CA + 31) / 32;          // 32-channel chunks of the reduction
  constexpr int TS = (CA >= 32) ? 4 : (CA + 7) / 8;   // 8-channel steps per chunk
  __shared__ __attribute__((aligned(16))) float acc[OS_ROWS * ST];
  __shared__ int32_t lists[OS_WAVES][64];
  __shared__ int turn;
  __shared__ double os_part[OS_WAVES][2][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int p = lane & 15, g = lane >> 4;
  int32_t *list = lists[wave];
  const int64_t r0 = (int64_t)blockIdx.x * OS_ROWS;

  for (int e = tid; e < OS_ROWS * ST / 4; e += 64 * OS_WAVES) ((float4 *)acc)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid == 0) turn = 0;
  __syncthreads();

  const bool row_ok = r0 + lane < n_out;
  const int64_t my = row_ok ? r0 + lane : n_out - 1;
  auto nbr_at = [&](int k) { return nbr[(int64_t)(flip ? kvol - 1 - k : k) * n_out + my]; };
  // The turn lives in LDS and orders LDS traffic only.  The LDS unit executes a wave's instructions in order, so a wave that read
  // turn == k issues its accumulator reads after that read, and its accumulator writes precede its store of k + 1: RELAXED accesses
  // plus compiler barriers are enough.  A workgroup-scope RELEASE would also drain the wave's outstanding GLOBAL loads
  // (s_waitcnt vmcnt(0)) -- the prefetched neighbour indices of its next offset.
  auto wait_turn = [&](int k) {
    int spins = 0;
    while (__hip_atomic_load(&turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != k && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };
  auto pass_turn = [&](int k) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(&turn, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  int32_t v_next = wave < kvol ? nbr_at(wave) : -1;
  for (int k = wave; k < kvol; k += OS_WAVES) {
    const int32_t v = v_next;
    if (k + OS_WAVES < kvol) v_next = nbr_at(k + OS_WAVES);
    const bool valid = row_ok && v >= 0 && v < rows_a;
    const unsigned long long mask = __ballot(valid);
    const int cnt = __popcll(mask);
    if (cnt == 0) {   // wave-uniform
      wait_turn(k);
      pass_turn(k);
      continue;
    }
    // W[k] of this offset, in registers for all its tiles.  Issued BEFORE the gathers (and independent of the neighbour indices), so
    // the two latencies overlap: lane (column p, group g) needs rows cb + (g>>1) and cb + (g>>1) + 2 of every 8-channel step.
    const float *Wk = W + (int64_t)k * CA * CO;
    float w1[NCH][TS][CO16], w2[NCH][TS][CO16];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const int cb = ch * 32 + 8 * t + 4 * (g & 1);   // this lane group's 4-channel base
#pragma unroll
        for (int j = 0; j < CO16; ++j) {
          w1[ch][t][j] = w2[ch][t][j] = 0.f;
          if (cb + 4 <= CA) {
            if (WT) {   // W[k] stored (CO, CA): four consecutive reduction channels in one 16-byte load
              const float4 wv = *(const float4 *)(Wk + (int64_t)(16 * j + p) * CA + cb);
              w1[ch][t][j] = (g >> 1) ? wv.y : wv.x;
              w2[ch][t][j] = (g >> 1) ? wv.w : wv.z;
            } else {    // W[k] stored (CA, CO): 16 lanes read 64 contiguous bytes of a row
              w1[ch][t][j] = Wk[(int64_t)(cb + (g >> 1)) * CO + 16 * j + p];
              w2[ch][t][j] = Wk[(int64_t)(cb + (g >> 1) + 2) * CO + 16 * j + p];
            }
          }
        }
      }
    if (valid) {
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
      list[rank] = (v << 6) | lane;
    }
    __builtin_amdgcn_wave_barrier();
    bool have_turn = false;
    for (int t0 = 0; t0 < cnt; t0 += 16) {
      const bool ok = t0 + p < cnt;
      const int32_t ent = list[ok ? t0 + p : 0];   // entry 0 exists (cnt >= 1): a padded lane gathers a real row and is never stored
      const float *arow = A + (int64_t)(ent >> 6) * CA;
      float4 a[NCH][TS];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const int col = ch * 32 + 8 * t + 4 * (g & 1);
          a[ch][t] = (col + 4 <= CA) ? *(const float4 *)(arow + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      os_f32x4 c[CO16];
#pragma unroll
      for (int j = 0; j < CO16; ++j) c[j] = (os_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const float a1 = (g >> 1) ? a[ch][t].y : a[ch][t].x;    // channel cb + (g>>1)
          const float a2 = (g >> 1) ? a[ch][t].w : a[ch][t].z;    // channel cb + (g>>1) + 2
#pragma unroll
          for (int j = 0; j < CO16; ++j) {
            c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[ch][t][j], a1, c[j], 0, 0, 0);
            c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[ch][t][j], a2, c[j], 0, 0, 0);
          }
        }
      if (!have_turn) {   // the first tile's products are done: now wait for the offsets before this one
        wait_turn(k);
        have_turn = true;
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
    pass_turn(k);
  }
  __syncthreads();   // every offset has been added

  for (int e = tid; e < OS_ROWS * (CO / 4); e += 64 * OS_WAVES) {   // the block's 64 x CO rows, written once
    const int row = e / (CO / 4), c4 = e - row * (CO / 4);
    if (r0 + row < n_out) *(float4 *)(out + (r0 + row) * CO + 4 * c4) = *(const float4 *)(acc + row * ST + 4 * c4);
  }
  if (part == nullptr) return;   // launch-uniform

  // BatchNorm statistics of the block's rows: column sums and sums of squares in float64; a column's 64 rows are split over the
  // waves (consecutive rows each, in order), the slices are added in wave order, blocks in block order by the last block to
  // finish (ftx_lastblock.h).  Rows past n_out are zero rows.
  if (lane < CO) {
    double s0 = 0, s1 = 0;
    constexpr int RW = OS_ROWS / OS_WAVES;
    for (int r = RW * wave; r < RW * wave + RW; ++r) {
      const double x = (double)acc[r * ST + lane];
      s0 += x;
      s1 += x * x;
    }
    os_part[wave][0][lane] = s0;
    os_part[wave][1][lane] = s1;
  }
  __syncthreads();
  if (tid >= 256) return;   // the hand-over below is written for 256 threads; waves that have ended do not count in a barrier
  for (int e = tid; e < 2 * CO; e += 256) {
    const int which = e / CO, col = e - which * CO;
    double s = os_part[0][which][col];
#pragma unroll
    for (int w = 1; w < OS_WAVES; ++w) s += os_part[w][which][col];
    lb_store(&part[((int64_t)blockIdx.x * 2 + which) * CO + col], s);
  }
  __syncthreads();   // the statistics slices are dead: their LDS (8 KB) is the hand-over's scratch (256 + 2 CO doubles <= 3 KB)
  last_block_totals(part, (int)gridDim.x, CO, sc, &os_part[0][0][0], StoreTotals{part + (int64_t)gridDim.x * 2 * CO, CO});
}

namespace {
template <int CA, int CO16, bool WT>
int os_launch(unsigned grid, hipStream_t st, const float *A, int64_t rows_a, const int32_t *nbr, int64_t n_out, const float *W, int flip, int kvol,
              float *out, double *part, StreamScratch sc) {
  // 4 waves per block; 8 (512 threads, template argument OS_WAVES) measured the same (tools/probes/ostat_variants/README.md)
  spconv_ostat_kernel<CA, CO16, WT, 4><<<grid, 256, 0, st>>>(A, rows_a, nbr, n_out, W, flip, kvol, out, part, sc);
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
  const int64_t b = ceil_div(n_out < 1 ? 1 : n_out, (int64_t)OS_ROWS);
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
