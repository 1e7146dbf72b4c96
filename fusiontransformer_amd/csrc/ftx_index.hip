// Integer side of the sparse-voxel path: coordinate hashing, the HBM-resident
// open-addressing hash table, counting, sorted unique, stride-2 coordinate
// downsample, kernel-map (neighbour table) construction and trilinear weights.
// All of it is HBM-bound integer work: one thread per row / per (offset,row),
// 16-byte coordinate loads, coalesced int32 stores.
#include <cstring>
#include <cstdlib>
#include "ftx_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <string>

namespace ftx {
static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
}  // namespace ftx

using namespace ftx;

extern "C" int ftx_version(void) { return 100; }
extern "C" const char *ftx_last_error(void) { return ftx::g_err.c_str(); }

// ---------------------------------------------------------------- hashing
__global__ void hash_kernel(const int4 *__restrict__ coords, int64_t n, int64_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int4 c = coords[i];
    out[i] = fnv_hash4(c.x, c.y, c.z, c.w);
  }
}

extern "C" int ftx_hash(const int32_t *coords, int64_t n, int64_t *out, void *stream) {
  FTX_REQUIRE(n >= 0, "ftx_hash: n < 0");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(coords && out, "ftx_hash: null pointer");
  hash_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)coords, n, out);
  return check_launch("ftx_hash");
}

__global__ void hash_kernel_offsets(const int4 *__restrict__ coords, int64_t n, const int32_t *__restrict__ offsets, int k,
                                    int64_t *__restrict__ out) {
  int64_t total = n * k;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int kk = (int)(e / n);
    int64_t i = e - (int64_t)kk * n;
    int4 c = coords[i];
    out[e] = fnv_hash4(c.x + offsets[kk * 3 + 0], c.y + offsets[kk * 3 + 1], c.z + offsets[kk * 3 + 2], c.w);
  }
}

extern "C" int ftx_hash_kernel(const int32_t *coords, int64_t n, const int32_t *offsets, int32_t k, int64_t *out, void *stream) {
  FTX_REQUIRE(n >= 0 && k >= 0, "ftx_hash_kernel: negative size");
  if (n == 0 || k == 0) return FTX_OK;
  FTX_REQUIRE(coords && offsets && out, "ftx_hash_kernel: null pointer");
  hash_kernel_offsets<<<grid_for(n * k, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)coords, n, offsets, k, out);
  return check_launch("ftx_hash_kernel");
}

__global__ void floor_coords_kernel(const float4 *__restrict__ pc, int64_t n, int stride, int4 *__restrict__ out) {
  const float s = (float)stride;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float4 p = pc[i];
    int4 o;
    o.x = (int)floorf(p.x / s) * stride;
    o.y = (int)floorf(p.y / s) * stride;
    o.z = (int)floorf(p.z / s) * stride;
    o.w = (int)p.w;
    out[i] = o;
  }
}

extern "C" int ftx_floor_coords(const float *pc, int64_t n, int32_t stride, int32_t *out, void *stream) {
  FTX_REQUIRE(n >= 0 && stride >= 1, "ftx_floor_coords: bad size/stride");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(pc && out, "ftx_floor_coords: null pointer");
  floor_coords_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const float4 *)pc, n, stride, (int4 *)out);
  return check_launch("ftx_floor_coords");
}

// ---------------------------------------------------------------- hash table
extern "C" int64_t ftx_hashtable_capacity(int64_t n) {
  int64_t c = 64;
  while (c < 2 * n) c <<= 1;
  return c;
}

__global__ void table_init_kernel(int64_t *__restrict__ tk, int32_t *__restrict__ tv, int64_t cap) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < cap; i += (int64_t)gridDim.x * blockDim.x) {
    tk[i] = kEmptyKey;
    tv[i] = 0x7fffffff;
  }
}

__global__ void table_insert_kernel(const int64_t *__restrict__ keys, int64_t n, int64_t *__restrict__ tk, int32_t *__restrict__ tv,
                                    int64_t cap) {
  const uint64_t mask = (uint64_t)cap - 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t key = keys[i];
    uint64_t slot = slot_mix((uint64_t)key) & mask;
    for (int64_t probe = 0; probe < cap; ++probe) {
      unsigned long long prev = atomicCAS((unsigned long long *)&tk[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (prev == (unsigned long long)kEmptyKey || prev == (unsigned long long)key) {
        atomicMin(&tv[slot], (int32_t)i);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

__device__ inline int32_t table_lookup(int64_t key, const int64_t *__restrict__ tk, const int32_t *__restrict__ tv, uint64_t mask,
                                       int64_t cap) {
  uint64_t slot = slot_mix((uint64_t)key) & mask;
  for (int64_t probe = 0; probe < cap; ++probe) {
    int64_t cur = tk[slot];
    if (cur == key) return tv[slot];
    if (cur == kEmptyKey) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

__global__ void table_query_kernel(const int64_t *__restrict__ q, int64_t nq, const int64_t *__restrict__ tk,
                                   const int32_t *__restrict__ tv, int64_t cap, int32_t *__restrict__ out) {
  const uint64_t mask = (uint64_t)cap - 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = table_lookup(q[i], tk, tv, mask, cap);
}

static bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int ftx_hashtable_build(const int64_t *keys, int64_t n, int64_t *table_keys, int32_t *table_vals, int64_t capacity,
                                   void *stream) {
  FTX_REQUIRE(n >= 0, "ftx_hashtable_build: n < 0");
  FTX_REQUIRE(is_pow2(capacity) && capacity >= 2 * n && capacity >= 64, "ftx_hashtable_build: capacity %lld must be a power of two >= max(64, 2n)", (long long)capacity);
  FTX_REQUIRE(table_keys && table_vals && (keys || n == 0), "ftx_hashtable_build: null pointer");
  table_init_kernel<<<grid_for(capacity, 256), 256, 0, (hipStream_t)stream>>>(table_keys, table_vals, capacity);
  if (n > 0) table_insert_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(keys, n, table_keys, table_vals, capacity);
  return check_launch("ftx_hashtable_build");
}

extern "C" int ftx_hashtable_query(const int64_t *queries, int64_t nq, const int64_t *table_keys, const int32_t *table_vals,
                                   int64_t capacity, int32_t *out, void *stream) {
  FTX_REQUIRE(nq >= 0, "ftx_hashtable_query: nq < 0");
  if (nq == 0) return FTX_OK;
  FTX_REQUIRE(is_pow2(capacity), "ftx_hashtable_query: capacity must be a power of two");
  FTX_REQUIRE(queries && table_keys && table_vals && out, "ftx_hashtable_query: null pointer");
  table_query_kernel<<<grid_for(nq, 256), 256, 0, (hipStream_t)stream>>>(queries, nq, table_keys, table_vals, capacity, out);
  return check_launch("ftx_hashtable_query");
}

// ---------------------------------------------------------------- count
__global__ void count_kernel(const int32_t *__restrict__ idx, int64_t n, int32_t *__restrict__ counts, int64_t m) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t p = idx[i];
    if (p >= 0 && p < m) atomicAdd(&counts[p], 1);
  }
}

extern "C" int ftx_count(const int32_t *idx, int64_t n, int32_t *counts, int64_t m, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0, "ftx_count: negative size");
  if (m == 0) return FTX_OK;
  FTX_REQUIRE(counts && (idx || n == 0), "ftx_count: null pointer");
  if (hipMemsetAsync(counts, 0, sizeof(int32_t) * m, (hipStream_t)stream) != hipSuccess) return check_launch("ftx_count memset");
  if (n > 0) count_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(idx, n, counts, m);
  return check_launch("ftx_count");
}

// ---------------------------------------------------------------- sorted unique
__global__ void iota_kernel(int32_t *__restrict__ v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct UniqueLayout {
  size_t off_keys, off_vals_in, off_vals_sorted, off_tmp, tmp_bytes, total;
};

static int unique_layout(int64_t n, UniqueLayout *L) {
  size_t sort_bytes = 0, uniq_bytes = 0;
  int64_t *kp = nullptr;
  int32_t *vp = nullptr;
  hipError_t e1 = rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, (size_t)n, 0u, 60u);
  hipError_t e2 = rocprim::unique_by_key(nullptr, uniq_bytes, kp, vp, kp, vp, vp, (size_t)n);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    set_error("ftx_unique: rocprim size query failed");
    return FTX_ELAUNCH;
  }
  L->off_keys = 0;
  L->off_vals_in = align256(L->off_keys + sizeof(int64_t) * n);
  L->off_vals_sorted = align256(L->off_vals_in + sizeof(int32_t) * n);
  L->off_tmp = align256(L->off_vals_sorted + sizeof(int32_t) * n);
  L->tmp_bytes = sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes;
  L->total = align256(L->off_tmp + L->tmp_bytes);
  return FTX_OK;
}

extern "C" size_t ftx_unique_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  UniqueLayout L;
  if (unique_layout(n, &L) != FTX_OK) return 0;
  return L.total;
}

extern "C" int ftx_unique_sorted(const int64_t *keys, int64_t n, int64_t *uniq, int32_t *first_index, int32_t *n_unique,
                                 void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n >= 0, "ftx_unique_sorted: n < 0");
  FTX_REQUIRE(n_unique, "ftx_unique_sorted: null n_unique");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(n_unique, 0, sizeof(int32_t), st) != hipSuccess) return check_launch("ftx_unique memset");
    return FTX_OK;
  }
  FTX_REQUIRE(keys && uniq && first_index && workspace, "ftx_unique_sorted: null pointer");
  FTX_REQUIRE(n < 0x7fffffff, "ftx_unique_sorted: n too large for int32 rows");
  UniqueLayout L;
  int rc = unique_layout(n, &L);
  if (rc != FTX_OK) return rc;
  if (workspace_bytes < L.total) {
    set_error("ftx_unique_sorted: workspace %zu < required %zu", workspace_bytes, L.total);
    return FTX_EWORKSPACE;
  }
  char *ws = (char *)workspace;
  int64_t *keys_sorted = (int64_t *)(ws + L.off_keys);
  int32_t *vals_in = (int32_t *)(ws + L.off_vals_in);
  int32_t *vals_sorted = (int32_t *)(ws + L.off_vals_sorted);
  void *tmp = ws + L.off_tmp;
  iota_kernel<<<grid_for(n, 256), 256, 0, st>>>(vals_in, n);
  size_t tb = L.tmp_bytes;
  // hashes are 60-bit non-negative: sort bits [0,60).  Radix sort is stable, so equal
  // keys keep ascending row order and the first of each run is the first occurrence.
  if (rocprim::radix_sort_pairs(tmp, tb, keys, keys_sorted, vals_in, vals_sorted, (size_t)n, 0u, 60u, st) != hipSuccess) {
    set_error("ftx_unique_sorted: radix sort failed");
    return FTX_ELAUNCH;
  }
  tb = L.tmp_bytes;
  if (rocprim::unique_by_key(tmp, tb, keys_sorted, vals_sorted, uniq, first_index, n_unique, (size_t)n,
                             rocprim::equal_to<int64_t>(), st) != hipSuccess) {
    set_error("ftx_unique_sorted: unique_by_key failed");
    return FTX_ELAUNCH;
  }
  return check_launch("ftx_unique_sorted");
}

// rank[i] = position of queries[i] in sorted[0 .. *n_sorted) (ascending, unique), or -1 when absent: with `sorted` = the output of
// ftx_unique_sorted this is numpy.unique's `return_inverse` (torchsparse sparse_quantize(..., return_invs=True),
// data/semantic_kitti/semantic_kitti_dataloader.py:231) -- the count stays on the device, nothing is read back.
__global__ void sorted_rank_kernel(const int64_t *__restrict__ sorted, const int32_t *__restrict__ n_sorted, const int64_t *__restrict__ q,
                                   int64_t nq, int64_t cap, int32_t *__restrict__ rank) {
  int64_t m = *n_sorted;
  if (m > cap) m = cap;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t key = q[i];
    int64_t lo = 0, hi = m;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    rank[i] = (lo < m && sorted[lo] == key) ? (int32_t)lo : -1;
  }
}

extern "C" int ftx_sorted_rank(const int64_t *sorted, const int32_t *n_sorted, int64_t capacity, const int64_t *queries, int64_t nq, int32_t *rank,
                               void *stream) {
  FTX_REQUIRE(nq >= 0 && capacity >= 0 && capacity < 0x7fffffff, "ftx_sorted_rank: bad size");
  if (nq == 0) return FTX_OK;
  FTX_REQUIRE(sorted && n_sorted && queries && rank, "ftx_sorted_rank: null pointer");
  sorted_rank_kernel<<<grid_for(nq, 256), 256, 0, (hipStream_t)stream>>>(sorted, n_sorted, queries, nq, capacity, rank);
  return check_launch("ftx_sorted_rank");
}

// ---------------------------------------------------------------- 3-D augmentation of the dataset side (f1)
// out[i,:] = points[i,:] . R for a 3x3 R, with the rounding of `points.dot(rot_matrix)` in the reference
// (data/utils/augmentation_3d.py:41): the sgemm behind numpy's dot walks K = 3 with one fused multiply-add per step,
// t = x*r0j; t = fma(y, r1j, t); t = fma(z, r2j, t)  (checked against numpy on 600 000 values: 0 differences; a plain
// mul / add chain differs in 23 % of them, and a voxel index can change with the last bit).
struct Rot3 { float r[9]; };
__global__ void rotate_points_kernel(const float *__restrict__ p, int64_t n, Rot3 R, float *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float t = __fmul_rn(x, R.r[j]);
      t = __fmaf_rn(y, R.r[3 + j], t);
      t = __fmaf_rn(z, R.r[6 + j], t);
      out[3 * i + j] = t;
    }
  }
}

extern "C" int ftx_rotate_points(const float *points, int64_t n, const float *rot_host, float *out, void *stream) {
  FTX_REQUIRE(n >= 0, "ftx_rotate_points: n < 0");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(points && rot_host && out, "ftx_rotate_points: null pointer");
  Rot3 R;
  for (int i = 0; i < 9; ++i) R.r[i] = rot_host[i];
  rotate_points_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(points, n, R, out);
  return check_launch("ftx_rotate_points");
}

// ---------------------------------------------------------------- downsample / gather of coordinate rows
__device__ inline int floor_div(int a, int b) {
  int q = a / b;
  return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

__global__ void downsample_kernel(const int4 *__restrict__ c, int64_t n, int ratio, int4 *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int4 v = c[i];
    v.x = floor_div(v.x, ratio) * ratio;
    v.y = floor_div(v.y, ratio) * ratio;
    v.z = floor_div(v.z, ratio) * ratio;
    out[i] = v;
  }
}

extern "C" int ftx_downsample_coords(const int32_t *coords, int64_t n, int32_t ratio, int32_t *out, void *stream) {
  FTX_REQUIRE(n >= 0 && ratio >= 1, "ftx_downsample_coords: bad size/ratio");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(coords && out, "ftx_downsample_coords: null pointer");
  downsample_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)coords, n, ratio, (int4 *)out);
  return check_launch("ftx_downsample_coords");
}

__global__ void gather_coords_kernel(const int4 *__restrict__ src, const int32_t *__restrict__ index, int64_t n, int4 *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = src[index[i]];
}

extern "C" int ftx_gather_coords(const int32_t *src, const int32_t *index, int64_t n, int32_t *out, void *stream) {
  FTX_REQUIRE(n >= 0, "ftx_gather_coords: n < 0");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(src && index && out, "ftx_gather_coords: null pointer");
  gather_coords_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)src, index, n, (int4 *)out);
  return check_launch("ftx_gather_coords");
}

// ---------------------------------------------------------------- all U-Net levels in one pass
// The voxel sets of every stride the network visits (models/spvcnn.py:104-126: strides 1, 2, 4, 8, 16) are all functions of the POINTS'
// integer coordinates: level s = unique(floor_div(p, s) * s) in ascending hash order.  The lazy form -- level l+1 from level l, as
// torchsparse's spdownsample does -- is a chain of five (hash, sort, unique, read the count back) rounds; floor division composes
// (floor(floor(x/2)/2) = floor(x/4)), so here every level is hashed straight from the points, the L x N (level tag | hash) keys are sorted
// ONCE, and ONE host read returns all level sizes.  Same sets, same order, same coordinates as the chain (tested bit for bit).
constexpr int LV_MAX = 8;               // level tags live in bits 60..62 of the key (hashes are 60-bit)
struct LevelStrides { int32_t s[LV_MAX]; };

__global__ void levels_hash_kernel(const int4 *__restrict__ pts, int64_t n, LevelStrides st, int nl, int64_t *__restrict__ keys, int32_t *__restrict__ vals) {
  const int64_t total = n * nl;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(e / n);
    const int64_t i = e - (int64_t)l * n;
    const int4 c = pts[i];
    const int s = st.s[l];
    const int64_t h = fnv_hash4(floor_div(c.x, s) * s, floor_div(c.y, s) * s, floor_div(c.z, s) * s, c.w);
    keys[e] = ((int64_t)l << 60) | h;
    vals[e] = (int32_t)i;
  }
}

// level_off[l] = first position of level l's run in the sorted unique keys (level_off[nl] = total), by binary search for the tag; every
// key then loses its tag.  The searches read the tagged keys, so they run in their own launch, before the strip pass.
__global__ void levels_offsets_kernel(const int64_t *__restrict__ uniq, const int32_t *__restrict__ n_unique, int nl, int32_t *__restrict__ level_off) {
  const int l = threadIdx.x;
  if (l > nl) return;
  const int64_t total = *n_unique;
  if (l == nl) { level_off[nl] = (int32_t)total; return; }
  const int64_t want = (int64_t)l << 60;      // first key >= (l << 60)
  int64_t lo = 0, hi = total;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (uniq[mid] < want) lo = mid + 1; else hi = mid;
  }
  level_off[l] = (int32_t)lo;
}
__global__ void levels_strip_kernel(int64_t *__restrict__ uniq, const int32_t *__restrict__ n_unique) {
  const int64_t total = *n_unique;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    uniq[i] &= 0x0FFFFFFFFFFFFFFFLL;
}

struct LevelsLayout {
  size_t off_keys, off_keys_sorted, off_vals_in, off_vals_sorted, off_count, off_tmp, tmp_bytes, total;
};
static int levels_layout(int64_t n, int nl, LevelsLayout *L) {
  const size_t m = (size_t)n * nl;
  size_t sort_bytes = 0, uniq_bytes = 0;
  int64_t *kp = nullptr;
  int32_t *vp = nullptr;
  hipError_t e1 = rocprim::radix_sort_pairs(nullptr, sort_bytes, kp, kp, vp, vp, m, 0u, 63u);
  hipError_t e2 = rocprim::unique_by_key(nullptr, uniq_bytes, kp, vp, kp, vp, vp, m);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    set_error("ftx_levels_unique: rocprim size query failed");
    return FTX_ELAUNCH;
  }
  L->off_keys = 0;
  L->off_keys_sorted = align256(L->off_keys + sizeof(int64_t) * m);
  L->off_vals_in = align256(L->off_keys_sorted + sizeof(int64_t) * m);
  L->off_vals_sorted = align256(L->off_vals_in + sizeof(int32_t) * m);
  L->off_count = align256(L->off_vals_sorted + sizeof(int32_t) * m);
  L->off_tmp = align256(L->off_count + 256);
  L->tmp_bytes = sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes;
  L->total = align256(L->off_tmp + L->tmp_bytes);
  return FTX_OK;
}

extern "C" size_t ftx_levels_workspace_bytes(int64_t n, int32_t n_levels) {
  if (n <= 0 || n_levels <= 0 || n_levels > LV_MAX) return 256;
  LevelsLayout L;
  if (levels_layout(n, n_levels, &L) != FTX_OK) return 0;
  return L.total;
}

extern "C" int ftx_levels_unique(const int32_t *points, int64_t n, const int32_t *strides, int32_t n_levels, int64_t *uniq, int32_t *first_index,
                                 int32_t *level_off, int64_t *sorted_keys, int32_t *order, void *workspace, size_t workspace_bytes, void *stream) {
  FTX_REQUIRE(n >= 0 && n_levels >= 1 && n_levels <= LV_MAX, "ftx_levels_unique: bad size (1..%d levels)", LV_MAX);
  FTX_REQUIRE(strides && level_off, "ftx_levels_unique: null strides / level_off");
  LevelStrides ls;
  for (int l = 0; l < LV_MAX; ++l) ls.s[l] = 1;
  for (int l = 0; l < n_levels; ++l) {
    FTX_REQUIRE(strides[l] >= 1, "ftx_levels_unique: strides must be >= 1");
    ls.s[l] = strides[l];
  }
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (hipMemsetAsync(level_off, 0, sizeof(int32_t) * (n_levels + 1), st) != hipSuccess) return check_launch("ftx_levels_unique memset");
    return FTX_OK;
  }
  FTX_REQUIRE(points && uniq && first_index && workspace, "ftx_levels_unique: null pointer");
  FTX_REQUIRE(n * n_levels < 0x7fffffff, "ftx_levels_unique: too many keys for int32 rows");
  LevelsLayout L;
  int rc = levels_layout(n, n_levels, &L);
  if (rc != FTX_OK) return rc;
  if (workspace_bytes < L.total) {
    set_error("ftx_levels_unique: workspace %zu < required %zu", workspace_bytes, L.total);
    return FTX_EWORKSPACE;
  }
  char *ws = (char *)workspace;
  // sorted_keys / order (optional outputs): the n_levels x n (tag | hash) keys in sorted order and the point of each -- level l owns
  // [l*n, (l+1)*n): its points sorted by voxel, stable, i.e. the sorted segments of spvoxelize at that stride (ftx_level_segments)
  int64_t *keys = (int64_t *)(ws + L.off_keys), *keys_sorted = sorted_keys ? sorted_keys : (int64_t *)(ws + L.off_keys_sorted);
  int32_t *vals_in = (int32_t *)(ws + L.off_vals_in), *vals_sorted = order ? order : (int32_t *)(ws + L.off_vals_sorted);
  int32_t *count = (int32_t *)(ws + L.off_count);
  void *tmp = ws + L.off_tmp;
  const size_t m = (size_t)n * n_levels;
  levels_hash_kernel<<<grid_for((int64_t)m, 256), 256, 0, st>>>((const int4 *)points, n, ls, n_levels, keys, vals_in);
  size_t tb = L.tmp_bytes;
  // stable radix sort over (tag | hash): equal keys keep ascending point order, so the first of each run is the first occurrence
  if (rocprim::radix_sort_pairs(tmp, tb, keys, keys_sorted, vals_in, vals_sorted, m, 0u, 63u, st) != hipSuccess) {
    set_error("ftx_levels_unique: radix sort failed");
    return FTX_ELAUNCH;
  }
  tb = L.tmp_bytes;
  if (rocprim::unique_by_key(tmp, tb, keys_sorted, vals_sorted, uniq, first_index, count, m, rocprim::equal_to<int64_t>(), st) != hipSuccess) {
    set_error("ftx_levels_unique: unique_by_key failed");
    return FTX_ELAUNCH;
  }
  levels_offsets_kernel<<<1, 64, 0, st>>>(uniq, count, n_levels, level_off);
  levels_strip_kernel<<<grid_for((int64_t)m, 256), 256, 0, st>>>(uniq, count);
  return check_launch("ftx_levels_unique");
}

// seg_off[v] = first position, among the n sorted keys of ONE level (tag | hash, ascending), of voxel v's run: a lower bound per voxel.
// With `order` = that level's slice of ftx_levels_unique's sorted points this is exactly what ftx_segment_build(idx_query, n, m) returns
// for the point -> voxel index of the level, without sorting anything again.
__global__ void level_segments_kernel(const int64_t *__restrict__ sorted_keys, int64_t n, const int64_t *__restrict__ uniq, int64_t m, int64_t tag,
                                      int32_t *__restrict__ seg_off) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v <= m; v += (int64_t)gridDim.x * blockDim.x) {
    if (v == m) { seg_off[m] = (int32_t)n; continue; }
    const int64_t want = (tag << 60) | uniq[v];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    seg_off[v] = (int32_t)lo;
  }
}
extern "C" int ftx_level_segments(const int64_t *sorted_keys, int64_t n, const int64_t *uniq, int64_t m, int32_t level, int32_t *seg_off, void *stream) {
  FTX_REQUIRE(n >= 0 && m >= 0 && level >= 0 && level < LV_MAX, "ftx_level_segments: bad size / level");
  FTX_REQUIRE(seg_off && (m == 0 || (sorted_keys && uniq)), "ftx_level_segments: null pointer");
  level_segments_kernel<<<grid_for(m + 1, 256), 256, 0, (hipStream_t)stream>>>(sorted_keys, n, uniq, m, (int64_t)level, seg_off);
  return check_launch("ftx_level_segments");
}

// coords of one level: out[r] = floor_div(points[first[r]], stride) * stride (batch column kept): the rows of that level in hash order
__global__ void level_coords_kernel(const int4 *__restrict__ pts, const int32_t *__restrict__ first, int64_t n, int stride, int4 *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int4 v = pts[first[i]];
    v.x = floor_div(v.x, stride) * stride;
    v.y = floor_div(v.y, stride) * stride;
    v.z = floor_div(v.z, stride) * stride;
    out[i] = v;
  }
}
extern "C" int ftx_level_coords(const int32_t *points, const int32_t *first_index, int64_t n, int32_t stride, int32_t *out, void *stream) {
  FTX_REQUIRE(n >= 0 && stride >= 1, "ftx_level_coords: bad size / stride");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(points && first_index && out, "ftx_level_coords: null pointer");
  level_coords_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)points, first_index, n, stride, (int4 *)out);
  return check_launch("ftx_level_coords");
}

// ---------------------------------------------------------------- kernel maps
// One thread per (offset k, output row o), o fastest: coordinate reads are 16-byte
// and re-served from L2 across the K passes, the table probes are the random part.
__global__ void kernel_map_kernel(const int4 *__restrict__ oc, int64_t n_out, const int32_t *__restrict__ offsets, int k,
                                  const int64_t *__restrict__ tk, const int32_t *__restrict__ tv, int64_t cap,
                                  int32_t *__restrict__ nbr) {
  const uint64_t mask = (uint64_t)cap - 1;
  const int64_t total = n_out * k;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int kk = (int)(e / n_out);
    int64_t o = e - (int64_t)kk * n_out;
    int4 c = oc[o];
    int64_t h = fnv_hash4(c.x + offsets[kk * 3 + 0], c.y + offsets[kk * 3 + 1], c.z + offsets[kk * 3 + 2], c.w);
    nbr[e] = table_lookup(h, tk, tv, mask, cap);
  }
}

extern "C" int ftx_kernel_map_build(const int32_t *out_coords, int64_t n_out, const int32_t *offsets, int32_t k,
                                    const int64_t *table_keys, const int32_t *table_vals, int64_t capacity, int32_t *nbr, void *stream) {
  FTX_REQUIRE(n_out >= 0 && k >= 1, "ftx_kernel_map_build: bad size");
  if (n_out == 0) return FTX_OK;
  FTX_REQUIRE(is_pow2(capacity), "ftx_kernel_map_build: capacity must be a power of two");
  FTX_REQUIRE(out_coords && offsets && table_keys && table_vals && nbr, "ftx_kernel_map_build: null pointer");
  kernel_map_kernel<<<grid_for(n_out * k, 256), 256, 0, (hipStream_t)stream>>>((const int4 *)out_coords, n_out, offsets, k, table_keys,
                                                                                table_vals, capacity, nbr);
  return check_launch("ftx_kernel_map_build");
}

// ---------------------------------------------------------------- trilinear weights
// float32 arithmetic in upstream calc_ti_weights' operation order; corner order (bx,by,bz) with z fastest
// (= KernelRegion(2, s, 1) offsets).
__global__ void trilinear_kernel(const float4 *__restrict__ pc, const int32_t *__restrict__ idx, int64_t n, int scale,
                                 float *__restrict__ w) {
  const float s = (float)scale;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float4 p = pc[i];
    float fx, fy, fz;
    if (scale != 1) {
      fx = floorf(p.x / s) * s; fy = floorf(p.y / s) * s; fz = floorf(p.z / s) * s;
    } else {
      fx = floorf(p.x); fy = floorf(p.y); fz = floorf(p.z);
    }
    float cx = fx + s, cy = fy + s, cz = fz + s;
    // float32 throughout, in upstream's operation order (torchsparse calc_ti_weights works in the dtype of `pc`, float32):
    // three-factor product, division by scale^3, zero for absent corners, sum of the 8, division by (sum + 1e-8)
    const float lo[3] = {p.x - fx, p.y - fy, p.z - fz};
    const float hi[3] = {cx - p.x, cy - p.y, cz - p.z};
    float ws[8];
    float sum = 0.f;
    const float inv = s * s * s;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
      float v = ((bx ? lo[0] : hi[0]) * (by ? lo[1] : hi[1])) * (bz ? lo[2] : hi[2]);
      if (scale != 1) v = v / inv;
      if (idx[i * 8 + c] < 0) v = 0.f;
      ws[c] = v;
      sum += v;
    }
    const float den = sum + 1e-8f;
#pragma unroll
    for (int c = 0; c < 8; ++c) w[i * 8 + c] = ws[c] / den;
  }
}

extern "C" int ftx_trilinear_weights(const float *pc, const int32_t *idx, int64_t n, int32_t scale, float *weights, void *stream) {
  FTX_REQUIRE(n >= 0 && scale >= 1, "ftx_trilinear_weights: bad size/scale");
  if (n == 0) return FTX_OK;
  FTX_REQUIRE(pc && idx && weights, "ftx_trilinear_weights: null pointer");
  trilinear_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>((const float4 *)pc, idx, n, scale, weights);
  return check_launch("ftx_trilinear_weights");
}
