// Sparse-conv pair GEMM, LDS-DMA ring version (included by ftx_spconv.hip).
//
//   tmp[p, :] = A[gather[p], :] @ W[k(p)]        for the pairs p of a kernel map, grouped by offset k
//
// One persistent 512-thread block per CU walks tiles of 256 pairs x BN = 32 NT output channels.  The
// operands of a 32-deep reduction chunk (A rows: 32 KB, W chunk: BN x 128 B) are brought in by LDS-DMA
// (`global_load_lds_dwordx4`, no staging registers) into a ring of 3 stages, two chunks ahead of the MFMAs,
// across tile boundaries; a step costs one `s_waitcnt vmcnt(N)` + one barrier.  The two waves of a SIMD
// belong to the same block and work on the same stage, so one wave's fragment reads, DMA issue and
// epilogue stores overlap the other's MFMAs.  Why: measured with in-kernel timestamps, the register-staged
// 128-pair kernel spends 20 % of a tile's life in MFMAs per wave -- the rest is load issue / landing
// (latency under load is 1-1.5 us) and the tile prologue, and two or three co-resident blocks do not cover it.
//
// LDS images (a DMA writes 64 lanes x 16 B contiguously, so images cannot be padded; the swizzle is applied
// to the per-lane SOURCE address instead and again by the reader):
//   A stage   [256 pairs][8 x 16 B]   16-B piece c of row r is stored at position c ^ ((r >> 1) & 7)
//   W stage, W[k] stored (co, ca) (dgrad):   [BN][8 x 16 B], same swizzle, fragments by ds_read_b128
//   W stage, W[k] stored (ca, co) (forward): [32][BN] floats as they come, fragments by ds_read_b32
//   pair indices of a tile: 8 waves x 128 B, fetched by DMA two tiles ahead (4 slots)
#pragma once

namespace ring {

constexpr int BK = 32;
constexpr int IDX_SLOTS = 4;

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  // M0 carries the wave-uniform LDS destination; each active lane's 16 bytes land at lds_dst + lane * 16
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// wait until at most n of this wave's vector-memory operations (DMAs and stores, in issue order) are outstanding
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
#define FTX_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    FTX_W(0) FTX_W(1) FTX_W(2) FTX_W(3) FTX_W(4) FTX_W(5) FTX_W(6) FTX_W(7) FTX_W(8) FTX_W(9) FTX_W(10) FTX_W(11) FTX_W(12) FTX_W(13)
    FTX_W(14) FTX_W(15) FTX_W(16) FTX_W(17) FTX_W(18) FTX_W(19) FTX_W(20) FTX_W(21) FTX_W(22) FTX_W(23) FTX_W(24) FTX_W(25) FTX_W(26)
    FTX_W(27) FTX_W(28) FTX_W(29) FTX_W(30) FTX_W(31)
#undef FTX_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

struct Tile {
  int k;    // kernel offset (-1: no such tile)
  int p0;   // first pair
  int cnt;  // pairs in the tile
  int n0;   // first output column
};

template <int NT, int NW, int ST>
struct Layout {
  static constexpr int TILE = 32 * NW;                         // pairs per tile: NW waves x 32 pairs
  static constexpr int STAGES = ST;
  static constexpr int A_STAGE_BYTES = TILE * BK * 4;
  static constexpr int BN = 32 * NT;
  static constexpr int W_STAGE_BYTES = BN * BK * 4;
  static constexpr int STAGE_BYTES = A_STAGE_BYTES + W_STAGE_BYTES;
  static constexpr int IDX_OFF = STAGES * STAGE_BYTES;
  static constexpr int TAB_OFF = IDX_OFF + IDX_SLOTS * TILE * 4;
  static constexpr int LDS_BYTES = TAB_OFF + 2 * 65 * 4;
  static constexpr int W_INSTR = BN / 8;                       // 1-KiB DMA pieces per W chunk (4 NT)
  static constexpr int W_PER_WAVE = (W_INSTR + NW - 1) / NW;   // issued by each wave (surplus ones repeat a piece)
};

template <int NT, int NW, int ST>
__global__ __launch_bounds__(64 * NW) void pairs_gemm_ring_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ gather,
                                                              int64_t n_pairs, const float *__restrict__ W, int w_transposed,
                                                              const int32_t *__restrict__ koff, int ca, int co, int kvol,
                                                              float *__restrict__ tmp, const float *__restrict__ bias, int64_t n_dense, int ny) {
  using L = Layout<NT, NW, ST>;
  constexpr int BN = L::BN;
  constexpr int TILE = L::TILE, STAGES = L::STAGES, A_STAGE_BYTES = L::A_STAGE_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  int *s_idx = (int *)(smem + L::IDX_OFF);
  int *s_koff = (int *)(smem + L::TAB_OFF);
  int *s_excl = s_koff + 65;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const bool dense = (gather == nullptr);
  const int64_t n_rows_total = dense ? n_dense : n_pairs;

  if (!dense && tid < 64) {
    int lo = (lane <= kvol) ? koff[lane] : 0;
    int hi = (lane < kvol) ? koff[lane + 1] : lo;
    int nt = (hi - lo + TILE - 1) / TILE;
    int incl = nt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    s_koff[lane] = lo;
    s_excl[lane] = incl - nt;  // lanes >= kvol hold the total
    if (lane == 63) { s_koff[64] = lo; s_excl[64] = incl; }
  }
  __syncthreads();

  auto decode = [&](int id) {
    Tile d;
    d.k = -1; d.p0 = 0; d.cnt = 0;
    const int x = id / ny;
    d.n0 = (id - x * ny) * BN;
    if (dense) {
      int64_t left = n_dense - (int64_t)x * TILE;
      if (left > 0) { d.k = 0; d.p0 = x * TILE; d.cnt = left > TILE ? TILE : (int)left; }
    } else {
      bool mine = lane < kvol && x >= s_excl[lane] && x < s_excl[lane + 1];
      unsigned long long m = __ballot(mine);
      if (m != 0ull) {
        int kk = __ffsll((long long)m) - 1;
        d.k = kk;
        d.p0 = __builtin_amdgcn_readfirstlane(s_koff[kk] + (x - s_excl[kk]) * TILE);
        int left = __builtin_amdgcn_readfirstlane(s_koff[kk + 1]) - d.p0;
        d.cnt = left > TILE ? TILE : left;
      }
    }
    return d;
  };

  // ---- this lane's share of the DMAs -------------------------------------------------------------------------
  // A: wave w fills rows w*32 .. w*32+31 (its own MFMA rows) with 4 pieces of 8 rows
  const int a_rsub = lane >> 3;        // row inside a piece
  const int a_pos = lane & 7;          // 16-B position inside the stored row
  // pair indices: lanes 0..7 of wave w fetch the 32 indices of rows w*32.. of a tile into slot s
  auto issue_idx = [&](const Tile &d, int slot) {
    if (!dense && lane < 8) {
      int64_t p = (int64_t)d.p0 + wave * 32 + lane * 4;
      if (d.k < 0) p = 0;
      if (p > n_pairs - 4) p = n_pairs - 4;  // rows past the tile's end only need SOME valid index (masked at the store)
      glds16(gather + p, lds0 + L::IDX_OFF + (slot * TILE + wave * 32) * 4);
    }
  };
  // Clamped reads of a tile's indices; how far the clamp of issue_idx shifted them
  const float *a_src[4];   // row base + swizzled piece of this lane's 4 A rows of the loader's tile
  auto enter_tile = [&](const Tile &d, int slot) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = wave * 32 + u * 8 + a_rsub;       // tile row
      int64_t p = (int64_t)d.p0 + r;                   // pair (or dense row)
      int64_t src;
      if (dense) {
        src = p < n_rows_total ? p : n_rows_total - 1;
      } else {
        // slot holds gather[q .. q+3] per lane of issue_idx with q possibly clamped: recompute where pair p sits
        int64_t base = (int64_t)d.p0 + wave * 32;
        int64_t q = base + (int64_t)((u * 8 + a_rsub) >> 2) * 4;
        int64_t qc = q > n_pairs - 4 ? n_pairs - 4 : q;
        int64_t pc = p > n_pairs - 1 ? n_pairs - 1 : p;
        int off = (int)(pc - qc);                       // 0..3 normally; the clamp keeps it inside the 16 bytes fetched
        if (off < 0) off = 0;
        if (off > 3) off = 3;
        src = s_idx[slot * TILE + wave * 32 + ((u * 8 + a_rsub) >> 2) * 4 + off];
      }
      if (src < 0) src = 0;
      if (src >= rows_a) src = rows_a - 1;
      const int piece = a_pos ^ ((r >> 1) & 7);        // logical 16-B piece stored at position a_pos of row r
      a_src[u] = A + src * ca + piece * 4;
    }
  };
  auto issue_chunk = [&](const Tile &d, int c0, int stage) {
    const unsigned sa = lds0 + stage * L::STAGE_BYTES;
#pragma unroll
    for (int u = 0; u < 4; ++u) glds16(a_src[u] + c0, sa + (wave * 32 + u * 8) * 128);
    const unsigned sw = sa + A_STAGE_BYTES;
    const float *Wk = W + (int64_t)d.k * ca * co;
#pragma unroll
    for (int u = 0; u < L::W_PER_WAVE; ++u) {
      const int i = (u * NW + wave) % L::W_INSTR;       // 1-KiB piece of the W chunk
      const float *g;
      if (!w_transposed) {                             // image [32 k][BN]: W[k] rows are (ca, co)
        const int e = i * 64 + lane;                   // float4 index inside the image
        const int kk = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
        g = Wk + (int64_t)(c0 + kk) * co + d.n0 + n4;
      } else {                                         // image [BN n][8 pieces], swizzled like A: W[k] rows are (co, ca)
        const int n = i * 8 + (lane >> 3);
        const int piece = (lane & 7) ^ ((n >> 1) & 7);
        g = Wk + (int64_t)(d.n0 + n) * ca + c0 + piece * 4;
      }
      glds16(g, sw + i * 1024);
    }
  };
  constexpr int CHUNK_OPS = 4 + L::W_PER_WAVE;

  // ---- prologue: indices of the first two tiles, then the first two chunks ---------------------------------------
  const int G = gridDim.x;
  int seq = 0;                                  // tiles the loader has entered
  auto tile_id = [&](int n) { return (int)blockIdx.x + n * G; };
  Tile ld = decode(tile_id(0));
  if (ld.k < 0) return;
  issue_idx(ld, 0);
  issue_idx(decode(tile_id(1)), 1);
  wait_vm(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  issue_idx(decode(tile_id(2)), 2);
  enter_tile(ld, 0);
  int ld_c0 = 0;
  Tile d0 = ld;                                  // tile being multiplied
  int mseq = 0;
  int c0 = 0;
  int issued_prev;                               // ops of the batch issued one load step ago

  auto advance_loader = [&]() -> int {           // moves the loader one chunk on; returns the ops it issued for the idx ring
    int ops = 0;
    ld_c0 += BK;
    if (ld_c0 >= ca) {
      ld_c0 = 0;
      ++seq;
      ld = decode(tile_id(seq));
      if (ld.k >= 0) {
        enter_tile(ld, seq & (IDX_SLOTS - 1));
        issue_idx(decode(tile_id(seq + 2)), (seq + 2) & (IDX_SLOTS - 1));
        ops = dense ? 0 : 1;
      }
    }
    return ops;
  };

  // the loader runs LA = STAGES - 1 chunks ahead of the MFMAs: load steps 0 .. LA-1 go out here
  constexpr int LA = STAGES - 1;
  static_assert(LA == 1 || LA == 2, "the vmcnt bookkeeping below covers one or two chunks of lookahead");
  issue_chunk(ld, 0, 0);
  issued_prev = 0;                               // ops issued after the batch that the next wait retires
  if (LA == 2) {
    int ops1 = advance_loader();
    if (ld.k >= 0) { issue_chunk(ld, ld_c0, 1); ops1 += CHUNK_OPS; }
    issued_prev = ops1;                          // idx(tile 2) went out before chunk 0 and is covered by the first wait
  }

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;

  // fragment addresses: row l31 of the wave's 32 rows, piece (2t + half) ^ ((l31 >> 1) & 7)
  const int fsw = (l31 >> 1) & 7;
  int frag_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) frag_off[t] = l31 * 128 + (((2 * t + half) ^ fsw) << 4);

  Tile done;            // tile whose accumulators wait for their stores
  bool pending_store = false;
  done.k = -1; done.p0 = 0; done.cnt = 0; done.n0 = 0;

  for (int it = 0;; ++it) {
    // (1) the stage of this step has landed for every wave; the stage multiplied last step is free
    wait_vm(issued_prev);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int issued = 0;

    // (2) stores of the tile finished last step (its accumulators are still in place), then clear them
    if (pending_store) {
      const int row = wave * 32 + l31;
      if (row < done.cnt) {
        float *dst = tmp + (int64_t)(done.p0 + row) * co;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = done.n0 + j * 32 + 8 * q + 4 * half;
            float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
            if (bias) {
              const float4 bv = *(const float4 *)&bias[col];
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            *(float4 *)&dst[col] = v;
          }
      }
      // rows past the tile's end issue no store: their count is not uniform inside the wave, so wait for the worst case
      issued += 4 * NT;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
      pending_store = false;
    }
    if (d0.k < 0) break;

    // (3) DMAs LA chunks ahead, into the stage that was multiplied last step
    if (ld.k >= 0) {
      issued += advance_loader();
      if (ld.k >= 0) { issue_chunk(ld, ld_c0, (it + LA) % STAGES); issued += CHUNK_OPS; }
    }
    issued_prev = LA == 2 ? issued : 0;   // with one chunk of lookahead nothing younger than the awaited batch exists

    // (4) MFMAs of this step: W is the row operand, the accumulator holds (channel, pair)
    {
      const char *sa = smem + (it % STAGES) * L::STAGE_BYTES;
      const char *sw = sa + A_STAGE_BYTES;
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

    // (5) bookkeeping of the multiplied tile
    c0 += BK;
    if (c0 >= ca) {
      done = d0;
      pending_store = true;
      c0 = 0;
      ++mseq;
      d0 = decode(tile_id(mseq));
    }
  }
}

template <int NT, int NW, int ST>
static int launch_ring(int64_t tiles_x, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, int64_t n_pairs, const float *W,
                       int wT, const int32_t *koff, int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense) {
  using L = Layout<NT, NW, ST>;
  static bool configured = false;
  static int per_cu = 1;
  if (!configured) {
    if (hipFuncSetAttribute((const void *)pairs_gemm_ring_kernel<NT, NW, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, L::LDS_BYTES) != hipSuccess) return -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pairs_gemm_ring_kernel<NT, NW, ST>, 64 * NW, L::LDS_BYTES) != hipSuccess || per_cu < 1) per_cu = 1;
    configured = true;
  }
  const int ny = co / L::BN;
  int64_t total = tiles_x * ny;
  int64_t g = (int64_t)device_cus() * per_cu;
  if (g > total) g = total;
  pairs_gemm_ring_kernel<NT, NW, ST><<<(unsigned)g, 64 * NW, L::LDS_BYTES, st>>>(A, rows_a, gather, n_pairs, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, ny);
  return 0;
}

}  // namespace ring
