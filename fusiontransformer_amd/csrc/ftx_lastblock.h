// "The last block to finish reduces": column totals of per-block float64 partial rows without a second launch.
//
// A statistics kernel (bn_partial_kernel, spconv_reduce_stats_kernel) leaves one row part[b][2][c] per block.  The separate finalize
// launch that used to sum them cost 5-9 us of launch + drain per BatchNorm and direction (109 launches per training step).  Here
// the blocks take tickets instead: the last block of each group of LB_GROUP blocks sums its group's rows in block order into
// gpart[group], and the last group to finish sums the group rows in group order and hands the two totals of every column to
// `fin`.  WHICH block does the summing depends on timing; the order of every sum does not, so the totals are bit-reproducible.
//
// Tickets live in a small buffer per (device, stream) -- the caller's (ftx_stream_scratch_attach) or, for a stream nobody attached one
// to, the library's: launches of one stream are serialised, and the block that uses a counter last puts it back to zero, so the buffer
// is clean between kernels.  A kernel that dies mid-flight leaves tickets behind: ftx_stream_scratch_reset(stream) clears them, and a
// dirty ticket is caught by the device-side assert in last_block_totals instead of producing stale totals silently.
#pragma once
#include <assert.h>
#include "ftx_common.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "ftx_lastblock.h relies on gfx9 (CDNA) memory-instruction semantics: sc1 write-through stores and vmcnt-counted store acknowledgements"
#endif

namespace ftx {

constexpr int LB_GROUP = 32;        // blocks per first-level group
constexpr int LB_MAX_GROUPS = 128;  // => at most 4096 blocks per launch
constexpr int LB_MAX_COLS = 512;    // channels; a row holds 2 * 512 doubles

struct StreamScratch {
  uint32_t *counters;   // [1 + LB_MAX_GROUPS], zero between kernels
  double *gpart;        // [LB_MAX_GROUPS][2][LB_MAX_COLS]
};
// Per-stream buffer, created (and zeroed on that stream) at first use; nullptr members + ftx error text on failure.
StreamScratch stream_scratch(hipStream_t st);

// Device-wide visibility WITHOUT cache maintenance.  A __threadfence() per block (release / acquire at agent scope) is a write-back
// plus an invalidate of the XCD's whole L2 on gfx950, and with 2 000 blocks per launch that made the statistics kernels 3.7x slower
// (measured: spconv_reduce_stats 21 -> 78 us).  Instead every value that crosses blocks -- the partial rows, the group rows, the
// tickets -- moves by RELAXED atomic stores / loads / adds at agent scope, which go through to the coherence point (sc1) and leave
// the caches alone; "my stores are done" is an explicit s_waitcnt vmcnt(0) before the barrier that precedes
// the block's ticket, and the ticket's returned value gates the loads of the block that sums.
//
// What each step relies on (gfx950 ISA as emitted -- the disassembly lines are in profiles/r03_lastblock_isa.txt):
//  * lb_store  -> global_store_dwordx2 ... sc1           an agent-scope store: written THROUGH the non-coherent per-XCD L2 to the memory
//                                                        side (MALL / HBM), where every XCD sees it; nothing is left dirty in a cache
//  * lb_stores_done -> s_waitcnt vmcnt(0) ; s_barrier    on gfx9 vmcnt counts stores as well as loads and a store is counted down when
//                                                        the memory side has ACKNOWLEDGED it, so after the barrier every partial row of
//                                                        the block is visible device-wide.  (This is the hardware meaning of a release at
//                                                        agent scope minus the L2 write-back, which has nothing to write back here because
//                                                        all cross-block data went out with sc1.)  Outside the HIP memory model; gfx9 only.
//  * lb_ticket -> global_atomic_add ... sc0 (returning)  an RMW atomic at agent scope: always executed at the device's coherence point
//                                                        (RMW atomics bypass the non-coherent caches by themselves; an sc1 bit on an
//                                                        atomic would mean SYSTEM scope), in ticket order: the block that reads gsize-1
//                                                        knows every other block of the group took its ticket AFTER its own stores were
//                                                        acknowledged
//  * lb_load   -> global_load_dwordx2 ... sc1            an agent-scope load: misses the XCD's L2 on purpose and reads the memory side,
//                                                        so it cannot return a stale line; issued after the ticket's return value
//                                                        (s_waitcnt + the branch on s_last) -- the acquire side
__device__ inline void lb_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double lb_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline uint32_t lb_ticket(uint32_t *p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void lb_stores_done() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores are acknowledged (a workgroup-scope fence emits no wait)
  __syncthreads();
}

// tot[j] = sum over rows r < nrows of src[r * c2 + j], for the c2 <= 1024 columns of a row, by a block of 256 threads.  The loads
// are write-through-coherent (lb_load) and each one waits ~1 us, so they are issued eight at a time, and when c2 <= 128 the rows of
// a column are split over 256 / c2 threads (interleaved slices, combined in slice order through `slices`).  The order of every sum
// is a function of (nrows, c2) only.  `slices` holds 256 doubles, `tot` c2; ends with a block barrier.
__device__ inline void lb_column_sums(const double *src, int nrows, int c2, double *slices, double *tot) {
  const int tid = threadIdx.x;
  auto strided_sum = [&](int j, int first, int step) {
    double s = 0;
    for (int base = first; base < nrows; base += 8 * step) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = base + u * step;
        v[u] = lb_load(&src[(int64_t)(r < nrows ? r : first) * c2 + j]);
        if (r >= nrows) v[u] = 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    return s;
  };
  if (c2 <= 128) {
    const int R = 256 / c2, j = tid % c2, r = tid / c2;
    if (r < R) slices[r * c2 + j] = strided_sum(j, r, R);
    __syncthreads();
    if (tid < c2) {
      double t = 0;
      for (int q = 0; q < R; ++q) t += slices[q * c2 + tid];
      tot[tid] = t;
    }
  } else {
    for (int j = tid; j < c2; j += 256) tot[j] = strided_sum(j, 0, 1);
  }
  __syncthreads();
}

// Call from EVERY thread of EVERY block of 256 threads (block-uniform control flow) after the block's own row part[blockIdx.x] has
// been written WITH lb_store.  `lds`: 256 + 2c doubles of shared memory the caller no longer needs.  fin(col, total0, total1) runs
// once per column, in the last block only; what it writes is for LATER kernels.
template <class Fin>
__device__ inline void last_block_totals(const double *part, int nb, int c, StreamScratch sc, double *lds, Fin fin) {
  __shared__ int s_last;
  const int tid = threadIdx.x, c2 = 2 * c;
  const int group = blockIdx.x / LB_GROUP, ngroups = (nb + LB_GROUP - 1) / LB_GROUP;
  const int g0 = group * LB_GROUP;
  const int gsize = nb - g0 < LB_GROUP ? nb - g0 : LB_GROUP;
  double *slices = lds, *tot = lds + 256;
  lb_stores_done();
  if (tid == 0) {
    const uint32_t ticket = lb_ticket(&sc.counters[1 + group]);
    assert(ticket < (uint32_t)gsize && "ftx: dirty BatchNorm ticket (a kernel died on this stream? call ftx_stream_scratch_reset)");
    s_last = ticket == (uint32_t)gsize - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  lb_column_sums(part + (int64_t)g0 * c2, gsize, c2, slices, tot);
  for (int j = tid; j < c2; j += 256) lb_store(&sc.gpart[(int64_t)group * c2 + j], tot[j]);
  lb_stores_done();
  if (tid == 0) {
    __hip_atomic_store(&sc.counters[1 + group], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t ticket = lb_ticket(&sc.counters[0]);
    assert(ticket < (uint32_t)ngroups && "ftx: dirty BatchNorm group ticket (call ftx_stream_scratch_reset)");
    s_last = ticket == (uint32_t)ngroups - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(&sc.counters[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  lb_column_sums(sc.gpart, ngroups, c2, slices, tot);
  for (int col = tid; col < c; col += 256) fin(col, tot[col], tot[c + col]);
}

// Stores the two totals of every column as a row [2][c] (the form the BatchNorm apply kernels read).
struct StoreTotals {
  double *totals;
  int c;
  __device__ void operator()(int col, double s, double ss) const {
    totals[col] = s;
    totals[c + col] = ss;
  }
};

}  // namespace ftx
