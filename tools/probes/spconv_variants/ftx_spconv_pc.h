// Pair GEMM as a persistent producer / consumer kernel (included by ftx_spconv.hip).
//
// Why: the tile kernels above are bound by per-tile latency (index load -> gather -> four steps that each wait for a gather issued one
// step earlier -> epilogue) at 2-4 resident waves per SIMD; a wave issues MFMAs for ~23 % of a tile's life and the matrix pipe is busy
// 54-65 % (profiles/r02_pmc_spconv_sq_waits.txt).  Here the roles are separated inside one 768-thread workgroup per CU:
//   waves 0-3   LOADERS: gather the operands of 32-deep chunks straight into an LDS ring (global_load_lds_dwordx4, no registers),
//               running up to RING chunks -- i.e. up to a whole tile -- ahead of the matrix waves, across tile boundaries;
//   waves 4-7   CONSUMER group 0, waves 8-11 CONSUMER group 1: each group takes every other tile of the workgroup's tile list,
//               reads fragments from the ring, issues the MFMAs and stores its tile; while one group is in its epilogue or waits for
//               a chunk, the other one owns the SIMD's matrix pipe (a workgroup's waves land on SIMDs cyclically, so every SIMD
//               hosts one loader and one wave of each group).
// No workgroup barrier after the prologue: a ring slot carries a FULL counter (each loader wave adds 1 when its part of the slot has
// landed: its own `s_waitcnt vmcnt`) and a FREE counter (each wave of the consuming group adds 1 after its last fragment read of the
// slot).  Counters only grow (slot s is full for its n-th fill when FULL[s] == 4 (n+1), free for it when FREE[s] == 4 n), every
// poll is bounded, and all waves walk the same tile list (tile t of the launch belongs to workgroup t mod gridDim.x), so the kernel
// drains whatever the data is.
//
// Arithmetic, operand images, swizzle and the 16-byte epilogue are those of ftx_spconv_dma.h / pairs_gemm_kernel: exact-f32 MFMA,
// bit-identical results.  Whole chunks and whole column tiles only (ca % 32 == 0, co % (32 NT) == 0); everything else stays on
// pairs_gemm_kernel.  ftx_spconv_set_gemm_variant(2) / FTX_GEMM_PC=1 selects it.
#pragma once

namespace pc {

constexpr int TILE = 128;
constexpr int BK = 32;
constexpr int A_BYTES = TILE * BK * 4;
constexpr int RING = 4;
constexpr int TR = 16;     // tile descriptors kept in LDS: a loader is never more than 2 RING + 1 tiles ahead of a consumer's epilogue
constexpr int SPIN_LIMIT = 1 << 17;   // ~25 M cycles: three orders of magnitude above any legitimate wait

// The hand-over counters live in LDS and must be read as LDS (ds_read), never through a generic pointer: a flat load counts on vmcnt as
// well, so a poll would drain every LDS-DMA in flight.  Hence macros over the __shared__ arrays instead of functions taking pointers.
#define FTX_PC_WAIT(arr, idx, target)                                   \
  do {                                                                  \
    int spins_ = 0;                                                     \
    while (__hip_atomic_load(&(arr)[(idx)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (target)) { \
      __builtin_amdgcn_s_sleep(1);                                      \
      if (++spins_ > SPIN_LIMIT) break; /* bounded: never hang the GPU */ \
    }                                                                   \
  } while (0)

// tile index (over all offsets, then column tiles) -> offset, first pair, pair count, column tile.  Every wave computes it by itself
// from a 64-lane scan of the per-offset tile counts; returns false past the last tile.
struct TileInfo {
  int k, p0, cnt, ct;
};
// kc / kb: this lane's offset (lane < kvol) pair count and first pair, loaded ONCE per wave: the lookup itself is shuffles only, so a
// loader can look a tile ahead without touching memory.
__device__ __forceinline__ bool tile_lookup(int t, int kc, int kb, bool dense, int col_tiles, int64_t n_dense, int lane, TileInfo &ti) {
  const int pt = t / col_tiles;
  ti.ct = t - pt * col_tiles;
  if (dense) {
    const int64_t left = n_dense - (int64_t)pt * TILE;
    if (left <= 0) return false;
    ti.k = 0;
    ti.p0 = pt * TILE;
    ti.cnt = left > TILE ? TILE : (int)left;
    return true;
  }
  int nt = (kc + TILE - 1) / TILE;
  int incl = nt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  const int excl = incl - nt;
  const bool mine = pt >= excl && pt < incl;   // lanes >= kvol have nt == 0: never "mine"
  const unsigned long long m = __ballot(mine);
  if (m == 0ull) return false;
  const int src = __ffsll((long long)m) - 1;
  const int tt = pt - __shfl(excl, src, 64);
  const int cc = __shfl(kc, src, 64);
  const int left = cc - tt * TILE;
  ti.k = src;
  ti.p0 = __shfl(kb, src, 64) + tt * TILE;
  ti.cnt = left > TILE ? TILE : left;
  return true;
}

__device__ __forceinline__ int tile_total(int kc, bool dense, int col_tiles, int64_t n_dense) {
  if (dense) return (int)ceil_div(n_dense, (int64_t)TILE) * col_tiles;
  int nt = (kc + TILE - 1) / TILE;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nt += __shfl_xor(nt, off, 64);
  return nt * col_tiles;
}

// one dword per lane, global -> LDS at lds_dst + 4 * lane (index prefetch of the loaders: counted on vmcnt, no VGPR, no compiler-inserted wait)
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int NT>
__global__ __launch_bounds__(768) void pairs_gemm_pc_kernel(const float *__restrict__ A, int64_t rows_a, const int32_t *__restrict__ gather,
                                                            const float *__restrict__ W, int w_transposed, const int32_t *__restrict__ koff,
                                                            int ca, int co, int kvol, float *__restrict__ tmp, const float *__restrict__ bias,
                                                            int64_t n_dense, const int32_t *__restrict__ scatter, int64_t rows_out, int debug) {
  // debug (measurement aid, FTX_PC_DEBUG): 1 = consumers skip the MFMAs (memory + hand-over only), 2 = loaders skip the DMAs (compute +
  // hand-over + stores, results are garbage), 4 = bare consumer loop (no DMAs, no hand-over, NO STORES: a consumer that ignores the
  // hand-over must never store, its destination rows may not be written yet), 0 = the product
  constexpr int BN = 32 * NT;
  constexpr int W_BYTES = BN * BK * 4;
  constexpr int STAGE = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [RING stages][A image | W image]
  __shared__ unsigned s_full[RING], s_free[RING];
  // Per-tile descriptors written by the loaders (which have the tile's indices in hand anyway), read by the consumers: a consumer wave
  // never issues a global LOAD -- its tile lookup and destination rows come from here -- so nothing but the ring stands between two of
  // its MFMA phases except its own stores.  s_drow: destination row of every pair of the tile; -1 = no store; <= -2 encodes "store a
  // zero row at -2 - value" (a pair whose source index was out of range).
  __shared__ int s_desc[TR][4];
  __shared__ int s_drow[TR][TILE];
  // index prefetch of the loaders: [buffer][loader wave][0..31 gather index, 32..63 scatter index of the wave's 32 pairs], fetched by
  // LDS-DMA one tile ahead so that the loader loop contains no global load the compiler would wait for
  __shared__ __attribute__((aligned(256))) int s_idx[2][4][64];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

  const int tid = threadIdx.x;
  if (tid < RING) {
    s_full[tid] = 0;
    s_free[tid] = 0;
  }
  __syncthreads();   // the only workgroup barrier

  const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col_tiles = co / BN;
  const int steps = ca / BK;
  const bool dense = (koff == nullptr);
  int kc = 0, kb = 0;
  if (!dense && lane < kvol) {
    kb = koff[lane];
    kc = koff[lane + 1] - kb;
  }
  const int total = tile_total(kc, dense, col_tiles, n_dense);

  if (wave < 4) {
    // ------------------------------------------------------------------ loader wave `wave`: rows [32 wave, 32 wave + 32) of A, pieces wave, wave+4, ... of W
    const int w_step = w_transposed ? BK : BK * co;
    unsigned q = 0;   // chunks issued by this workgroup so far (same sequence in every wave)
    unsigned signalled = 0;
    constexpr int PER_CHUNK = 4 + NT;   // DMA instructions per chunk and loader wave
    const unsigned idx0 = (unsigned)(size_t)(__attribute__((address_space(3))) int *)&s_idx[0][0][0];
    auto fetch_idx = [&](const TileInfo &t, int buf) {
      // lanes 0..31: gather index of pair 32 wave + lane, lanes 32..63: its scatter index (rows past the tile's end re-read pair p0)
      if (gather == nullptr) return;
      const int r = wave * 32 + (lane & 31);
      const int64_t p = t.p0 + (r < t.cnt ? r : 0);
      const int32_t *src = (lane < 32 || scatter == nullptr) ? gather + p : scatter + p;
      glds4(src, idx0 + (unsigned)((buf * 4 + wave) * 64 * 4));
    };
    TileInfo ti, tn;
    bool have = (int)blockIdx.x < total && tile_lookup(blockIdx.x, kc, kb, dense, col_tiles, n_dense, lane, ti);
    if (have) {
      fetch_idx(ti, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int j = 0; have; ++j) {
      const int buf = j & 1;
      // look one tile ahead and start fetching its indices: they are older than every chunk DMA of tile j, so they have landed when the
      // counted wait at the top of the next trip has let all but the youngest chunk DMAs retire
      const int tnext = blockIdx.x + (j + 1) * gridDim.x;
      const bool have_next = tnext < total && tile_lookup(tnext, kc, kb, dense, col_tiles, n_dense, lane, tn);
      if (have_next) fetch_idx(tn, buf ^ 1);
      const int n0 = ti.ct * BN;
      const float *Wk = W + (int64_t)ti.k * ca * co;
      const float *a_src[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rl = u * 8 + (lane >> 3);            // row inside this wave's 32-row group
        const int r = wave * 32 + rl;
        int64_t row = gather ? (int64_t)s_idx[buf][wave][rl] : (int64_t)(ti.p0 + (r < ti.cnt ? r : 0));
        if (row < 0 || row >= rows_a) row = 0;
        const int piece = (lane & 7) ^ ((r >> 1) & 7);
        a_src[u] = A + row * ca + piece * 4;
      }
      const float *w_src[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int i = u * 4 + wave;
        if (!w_transposed) {
          const int e = i * 64 + lane;
          const int kk = e / (BN / 4), n4 = (e % (BN / 4)) * 4;
          w_src[u] = Wk + (int64_t)kk * co + n0 + n4;
        } else {
          const int n = i * 8 + (lane >> 3);
          const int piece = (lane & 7) ^ ((n >> 1) & 7);
          w_src[u] = Wk + (int64_t)(n0 + n) * ca + piece * 4;
        }
      }
      {
        const int ts = j % TR;
        if (wave == 0 && lane == 0) {
          s_desc[ts][0] = ti.k;
          s_desc[ts][1] = ti.p0;
          s_desc[ts][2] = ti.cnt;
          s_desc[ts][3] = n0;
        }
        if (lane < 32) {
          const int r = wave * 32 + lane;
          int d = -1;
          if (r < ti.cnt) {
            const int64_t p = ti.p0 + r;
            bool zero = false;
            int64_t dr = p;
            if (gather) {
              const int32_t sidx = s_idx[buf][wave][lane];
              zero = sidx < 0 || sidx >= rows_a;
              if (scatter) {
                dr = s_idx[buf][wave][32 + lane];
                if (dr < 0 || dr >= rows_out) dr = -1;
              }
            }
            d = dr < 0 ? -1 : (zero ? (int)(-2 - dr) : (int)dr);
          }
          s_drow[ts][r] = d;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // descriptor in LDS before this wave publishes the tile's first chunk
      }
      for (int c = 0; c < steps; ++c, ++q) {
        const unsigned s = q % RING, fill = q / RING;
        if (debug != 4 && __hip_atomic_load(&s_free[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * fill) {
          // about to wait for the consumers: first make every chunk already issued visible to them (they may be waiting for exactly those)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          for (; signalled < q; ++signalled)
            if (lane == 0) __hip_atomic_fetch_add(&s_full[signalled % RING], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          FTX_PC_WAIT(s_free, s, 4u * fill);
        }
        const unsigned sa = lds0 + s * STAGE;
        if (debug != 2 && debug != 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) dma::glds16(a_src[u] + c * BK, sa + (wave * 32 + u * 8) * 128);
#pragma unroll
          for (int u = 0; u < NT; ++u) dma::glds16(w_src[u] + (int64_t)c * w_step, sa + A_BYTES + (u * 4 + wave) * 1024);
        }
        // at most RING-1 chunks stay in flight: when RING are unpublished the oldest has landed -> publish it
        if (q + 1 - signalled >= (unsigned)RING) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK * (RING - 1)) : "memory");
          if (lane == 0) __hip_atomic_fetch_add(&s_full[signalled % RING], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ++signalled;
        }
      }
      // next tile: its indices were fetched before this tile's chunk DMAs; all but the youngest min(steps, RING-1) chunks' DMAs retired
      // means they have landed (vmcnt retires in order)
      have = have_next;
      ti = tn;
      if (have) {
        if (steps >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK * 3) : "memory");
        else if (steps == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK * 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_CHUNK * 1) : "memory");
      }
    }
    // drain: publish the chunks still in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; signalled < q; ++signalled)
      if (lane == 0) __hip_atomic_fetch_add(&s_full[signalled % RING], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }

  // -------------------------------------------------------------------- consumer wave: group g, rows [32 cw, 32 cw + 32) of the tile
  const int g = (wave - 4) >> 2, cw = (wave - 4) & 3;
  const int fsw = (l31 >> 1) & 7;
  int frag_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) frag_off[t] = l31 * 128 + (((2 * t + half) ^ fsw) << 4);

  for (int j = g; (int)(blockIdx.x + j * gridDim.x) < total; j += 2) {
    const int ts = j % TR;
    f32x16 acc[NT];
#pragma unroll
    for (int jj = 0; jj < NT; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;

    for (int c = 0; c < steps; ++c) {
      const unsigned q = (unsigned)j * steps + c;
      const unsigned s = q % RING, fill = q / RING;
      if (debug != 4) FTX_PC_WAIT(s_full, s, 4u * (fill + 1));   // debug 4: bare consumer loop (no hand-over, no stores, garbage operands)
      const char *sa = smem + s * STAGE;
      const char *sw = sa + A_BYTES;
      const char *ap = sa + cw * 32 * 128;
      float af[2][4], bf[2][NT][4];
      auto load_frag = [&](int buf, int t) {
        float4 a = *(const float4 *)(ap + frag_off[t]);
        af[buf][0] = a.x; af[buf][1] = a.y; af[buf][2] = a.z; af[buf][3] = a.w;
        if (w_transposed) {
#pragma unroll
          for (int jj = 0; jj < NT; ++jj) {
            float4 b = *(const float4 *)(sw + jj * 32 * 128 + frag_off[t]);
            bf[buf][jj][0] = b.x; bf[buf][jj][1] = b.y; bf[buf][jj][2] = b.z; bf[buf][jj][3] = b.w;
          }
        } else {
          const float *wp = (const float *)sw + (8 * t + 4 * half) * BN + l31;
#pragma unroll
          for (int jj = 0; jj < NT; ++jj)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) bf[buf][jj][ss] = wp[ss * BN + jj * 32];
        }
      };
      load_frag(0, 0);
#pragma unroll
      for (int t = 0; t < BK / 8; ++t) {
        if (t + 1 < BK / 8) {
          load_frag((t + 1) & 1, t + 1);
        } else {
          // the last fragments of this slot are in registers once lgkmcnt drains: hand the slot back before the last MFMAs
          if (debug != 4) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&s_free[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
        if (debug != 1) {
#pragma unroll
          for (int ss = 0; ss < 4; ++ss)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj)
              acc[jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[t & 1][jj][ss], af[t & 1][ss], acc[jj], 0, 0, 0);
        } else {
#pragma unroll
          for (int jj = 0; jj < NT; ++jj) acc[jj][0] += bf[t & 1][jj][0] + af[t & 1][0];
        }
      }
    }

    // the tile's first chunk was published after its descriptor was written: it is visible now
    const int n0 = s_desc[ts][3];
    const int d = (debug == 4) ? -1 : s_drow[ts][cw * 32 + l31];
    const bool zero = d <= -2;
    const int64_t drow = zero ? (int64_t)(-2 - d) : (int64_t)d;
    if (drow >= 0 && debug != 4) {
      float *dst = tmp + drow * co;
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int col = n0 + jj * 32 + 8 * qq + 4 * half;
          float4 v = make_float4(acc[jj][4 * qq], acc[jj][4 * qq + 1], acc[jj][4 * qq + 2], acc[jj][4 * qq + 3]);
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

template <int NT>
static int launch(unsigned blocks, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W, int wT, const int32_t *koff,
                  int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense, const int32_t *scatter, int64_t rows_out) {
  constexpr int LDS = RING * (A_BYTES + 32 * NT * BK * 4);
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void *)pairs_gemm_pc_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    configured = true;
  }
  static const int debug = getenv("FTX_PC_DEBUG") ? atoi(getenv("FTX_PC_DEBUG")) : 0;
  pairs_gemm_pc_kernel<NT><<<blocks, 768, LDS, st>>>(A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out, debug);
  return 0;
}

static int dispatch(int nt, unsigned blocks, hipStream_t st, const float *A, int64_t rows_a, const int32_t *gather, const float *W, int wT,
                    const int32_t *koff, int ca, int co, int kvol, float *tmp, const float *bias, int64_t n_dense, const int32_t *scatter,
                    int64_t rows_out) {
  switch (nt) {
    case 1: return launch<1>(blocks, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out);
    case 2: return launch<2>(blocks, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out);
    case 3: return launch<3>(blocks, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out);
    default: return launch<4>(blocks, st, A, rows_a, gather, W, wT, koff, ca, co, kvol, tmp, bias, n_dense, scatter, rows_out);
  }
}

}  // namespace pc
