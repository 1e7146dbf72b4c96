// Inner-loop ceilings of the fp32 MFMA on gfx950: registers only, with LDS fragment reads (b128 / b32), 1-2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float As[128 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 128 * 36; i += 256) { As[i] = (float)(i % 7) * 0.01f; Bs[i] = (float)(i % 5) * 0.02f; }
  __syncthreads();
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  float a0 = As[tid], b0 = Bs[tid];
  const float *ap = &As[(wave * 32 + l31) * 36 + 4 * half];
  const float *bp = &Bs[l31 * 36 + 4 * half];
  const float *bq = &Bs[(4 * half) * 100 + l31];
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // registers only
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[j], 0, 0, 0);
    } else if (MODE == 1 || MODE >= 3) {  // b128 for both operands
      if (MODE == 3 || MODE == 4) __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 a = *(const float4 *)(ap + 8 * t);
        float av[4] = {a.x, a.y, a.z, a.w};
        float bv[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j) { float4 b = *(const float4 *)(bp + j * 32 * 36 + 8 * t); bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w; }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][s], av[s], acc[j], 0, 0, 0);
      }
    } else {  // b128 A, b32 B
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 a = *(const float4 *)(ap + 8 * t);
        float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < NT && j < 3; ++j) { float b = bq[(8 * t + s) * 100 + j * 32]; acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, av[s], acc[j], 0, 0, 0); }
      }
    }
    if (MODE != 0) asm volatile("" ::: "memory");
    if (MODE >= 3) {
      // the phased structure of the conv kernel: stage stores, barrier, (MFMAs above), barrier
      if (MODE == 3 || MODE == 5) {
        float4 v = make_float4(a0 + it, b0, a0, b0);
#pragma unroll
        for (int p = 0; p < 4; ++p) *(float4 *)&As[(p * 32 + (tid >> 3)) * 36 + (tid & 7) * 4] = v;
#pragma unroll
        for (int p = 0; p < 3; ++p) *(float4 *)&Bs[(p * 32 + (tid >> 3)) * 36 + (tid & 7) * 4] = v;
      }
      if (MODE == 3 || MODE == 4) __syncthreads();
    }
  }
  float s = 0.f;
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) s += acc[j][g];
  out[blockIdx.x * 256 + tid] = s;
}

// tile-like launch: `grid` blocks of `iters` steps each (a conv tile is 4 steps), optional 16-byte epilogue stores of the
// accumulators into a big buffer (EPI) and a dependent prologue load chain (PRO: index -> row gather, as a tile's first chunk)
template <int NT, int EPI, int PRO>
__global__ __launch_bounds__(256) void ktile(float *out, const int *idx, const float *rows, int iters) {
  __shared__ __attribute__((aligned(16))) float As[128 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  float seed = 0.f;
  if (PRO) {
    int i0 = idx[(blockIdx.x * 256 + tid) & 0xFFFFF];
    float4 v = *(const float4 *)&rows[(size_t)(i0 & 0xFFFFF) * 32 + (tid & 7) * 4];
    seed = v.x + v.y;
  }
  for (int i = tid; i < 128 * 36; i += 256) { As[i] = (float)(i % 7) * 0.01f + seed; Bs[i] = (float)(i % 5) * 0.02f; }
  __syncthreads();
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  const float *ap = &As[(wave * 32 + l31) * 36 + 4 * half];
  const float *bp = &Bs[l31 * 36 + 4 * half];
  for (int it = 0; it < iters; ++it) {
    float4 v = make_float4(seed + it, seed, seed, seed);
#pragma unroll
    for (int p = 0; p < 4; ++p) *(float4 *)&As[(p * 32 + (tid >> 3)) * 36 + (tid & 7) * 4] = v;
#pragma unroll
    for (int p = 0; p < 3; ++p) *(float4 *)&Bs[(p * 32 + (tid >> 3)) * 36 + (tid & 7) * 4] = v;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 a = *(const float4 *)(ap + 8 * t);
      float av[4] = {a.x, a.y, a.z, a.w};
      float bv[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j) { float4 b = *(const float4 *)(bp + j * 32 * 36 + 8 * t); bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w; }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][s], av[s], acc[j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (EPI) {
    float *dst = out + ((size_t)blockIdx.x * 128 + wave * 32 + l31) * (32 * NT);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(float4 *)&dst[j * 32 + 8 * q + 4 * half] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
  } else {
    float s = 0.f;
    for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) s += acc[j][g];
    if (s == 12345.f) out[blockIdx.x * 256 + tid] = s;
  }
}

// the same with the conv kernel's per-step register-staged gathers: 4 row pieces (random rows of a 41 MB table, 128 B each) and
// NT pieces of a small weight table per thread and step, issued before the MFMAs of a step and stored to LDS at the next one
template <int NT, int REAL>
__global__ __launch_bounds__(256) void kgather(float *out, const int *idx, const float *rows, const float *wtab, int iters, int ca) {
  __shared__ __attribute__((aligned(16))) float As[128 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int arow = tid >> 3, acol = (tid & 7) * 4;
  int src[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) src[p] = idx[((size_t)blockIdx.x * 128 + p * 32 + arow) & 0xFFFFF];
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  float4 ra[4], rb[NT];
  auto load = [&](int c0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *(const float4 *)&rows[(size_t)src[p] * ca + c0 + acol];
#pragma unroll
    for (int q = 0; q < NT; ++q) rb[q] = *(const float4 *)&wtab[((size_t)(blockIdx.x % 27) * ca + c0 + (tid >> 3)) * (32 * NT) + q * 32 + (tid & 7) * 4];
  };
  const float *ap = &As[(wave * 32 + l31) * 36 + 4 * half];
  const float *bp = &Bs[l31 * 36 + 4 * half];
  load(0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) *(float4 *)&As[(p * 32 + arow) * 36 + acol] = ra[p];
#pragma unroll
    for (int q = 0; q < NT; ++q) *(float4 *)&Bs[(q * 32 + arow) * 36 + acol] = rb[q];
    __syncthreads();
    if (it + 1 < iters) load((it + 1) * 32);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 a = *(const float4 *)(ap + 8 * t);
      float av[4] = {a.x, a.y, a.z, a.w};
      float bv[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j) { float4 b = *(const float4 *)(bp + j * 32 * 36 + 8 * t); bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w; }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][s], av[s], acc[j], 0, 0, 0);
    }
    __syncthreads();
  }
  float *dst = out + ((size_t)blockIdx.x * 128 + wave * 32 + l31) * (32 * NT);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4 *)&dst[j * 32 + 8 * q + 4 * half] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
}

// LDS-DMA form of the same tile: the 7 pieces per thread go global -> LDS directly (global_load_lds_dwordx4) into the other of two
// unpadded, XOR-swizzled buffers while the MFMAs run on the current one; one vmcnt(0) + barrier per step, no staging registers.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int NT, int NBUF>
__global__ __launch_bounds__(256) void kdma(float *out, const int *idx, const float *rows, const float *wtab, int iters, int ca) {
  constexpr int A_BYTES = 128 * 128, B_BYTES = 32 * NT * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *asrc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = wave * 32 + u * 8 + (lane >> 3);
    const int piece = (lane & 7) ^ ((r >> 1) & 7);
    asrc[u] = rows + (size_t)idx[((size_t)blockIdx.x * 128 + r) & 0xFFFFF] * ca + piece * 4;
  }
  auto issue = [&](int c0, int buf) {
    const unsigned sa = lds0 + buf * STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) glds16(asrc[u] + c0, sa + (wave * 32 + u * 8) * 128);
#pragma unroll
    for (int u = 0; u < NT; ++u) {   // W image [n][8 pieces] swizzled like A: wave w fills rows u*32 + w*8 ..
      const int n = u * 32 + wave * 8 + (lane >> 3);
      const int piece = (lane & 7) ^ ((n >> 1) & 7);
      glds16(wtab + ((size_t)(blockIdx.x % 27) * 32 * NT + n) * ca + c0 + piece * 4, sa + A_BYTES + (u * 32 + wave * 8) * 128);
    }
  };
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  const int fsw = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) foff[t] = l31 * 128 + (((2 * t + half) ^ fsw) << 4);
  issue(0, 0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + 1 < iters) issue((it + 1) * 32, (it + 1) % NBUF);
    const char *sa = smem + (it % NBUF) * STAGE, *sb = sa + A_BYTES;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 a = *(const float4 *)(sa + wave * 32 * 128 + foff[t]);
      float av[4] = {a.x, a.y, a.z, a.w};
      float bv[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j) { float4 b = *(const float4 *)(sb + j * 32 * 128 + foff[t]); bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w; }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][s], av[s], acc[j], 0, 0, 0);
    }
    if (NBUF == 1) __builtin_amdgcn_s_barrier();
  }
  float *dst = out + ((size_t)blockIdx.x * 128 + wave * 32 + l31) * (32 * NT);
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4 *)&dst[j * 32 + 8 * q + 4 * half] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
}

// wgrad-like tile stream with LDS-DMA: per step 32 pairs, A rows (128 ch) and G rows (32 NT ch) gathered by index into
// unpadded images [32 pairs][channels]; dW tile (128 x 32 NT) accumulates over `iters` steps; fragments by ds_read_b32.
template <int NT>
__global__ __launch_bounds__(256) void kdma_wgrad(float *out, const int *idx_a, const int *idx_g, const float *rows_a, const float *rows_g, int iters) {
  constexpr int TM = 128, TN = 32 * NT, A_BYTES = 32 * TM * 4, G_BYTES = 32 * TN * 4, STAGE = A_BYTES + G_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A image: 16 pieces of 1 KiB = 2 pairs x 512 B each; wave w issues pieces w*4 .. w*4+3 (pairs w*8 .. w*8+7)
  // G image: TN*4*32/1024 = NT*4 pieces; wave w issues pieces w*NT .. (rows of TN*4 bytes, linear)
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  const size_t base = (size_t)blockIdx.x * iters * 32;
  auto issue = [&](int it, int buf) {
    const unsigned sa = lds0 + buf * STAGE;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int piece = wave * 4 + u;                       // 1 KiB = pairs 2*piece, 2*piece+1
      const int pr = piece * 2 + (lane >> 5);
      const int ia = idx_a[(base + (size_t)it * 32 + pr) & 0xFFFFF];
      glds16(rows_a + (size_t)ia * TM + (lane & 31) * 4, sa + piece * 1024);
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int e = (wave * NT + u) * 64 + lane;            // float4 index in the G image
      const int pr = e / (TN / 4), c4 = (e % (TN / 4)) * 4;
      const int ig = idx_g[(base + (size_t)it * 32 + pr) & 0xFFFFF];
      glds16(rows_g + (size_t)ig * TN + c4, sa + A_BYTES + (wave * NT + u) * 1024);
    }
  };
  issue(0, 0);
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it + 1 < iters) issue(it + 1, (it + 1) & 1);
    const float *as = (const float *)(smem + (it & 1) * STAGE), *gs = (const float *)(smem + (it & 1) * STAGE + A_BYTES);
#pragma unroll
    for (int kk2 = 0; kk2 < 16; ++kk2) {
      const int kk = 2 * kk2 + half;
      float a = as[kk * TM + wave * 32 + l31];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float b = gs[kk * TN + j * 32 + l31];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[j], 0, 0, 0);
      }
    }
  }
  float *dst = out + ((size_t)blockIdx.x * 128 + wave * 32 + l31) * TN;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4 *)&dst[j * 32 + 8 * q + 4 * half] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
}

template <int NT>
void run_wgrad(const char *name, int grid, int iters, int n_rows) {
  constexpr int TM = 128, TN = 32 * NT;
  float *out, *ra, *rg; int *ia, *ig;
  (void)hipMalloc(&out, sizeof(float) * (size_t)grid * 128 * TN);
  (void)hipMalloc(&ra, sizeof(float) * (size_t)n_rows * TM); (void)hipMalloc(&rg, sizeof(float) * (size_t)n_rows * TN);
  (void)hipMalloc(&ia, sizeof(int) * (1 << 20)); (void)hipMalloc(&ig, sizeof(int) * (1 << 20));
  std::vector<int> h(1 << 20), h2(1 << 20); unsigned x = 777u;
  for (size_t i = 0; i < h.size(); ++i) { x = x * 1664525u + 1013904223u; h[i] = (int)((x >> 8) % (unsigned)n_rows); h2[i] = (int)((i / 5) % (unsigned)n_rows); }
  (void)hipMemcpy(ia, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(ig, h2.data(), sizeof(int) * h2.size(), hipMemcpyHostToDevice);
  std::vector<float> hr((size_t)n_rows * TM);
  for (auto &v : hr) { x = x * 1664525u + 1013904223u; v = (x >> 9) / 8388608.0f - 1.0f; }
  (void)hipMemcpy(ra, hr.data(), sizeof(float) * hr.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(rg, hr.data(), sizeof(float) * (size_t)n_rows * TN, hipMemcpyHostToDevice);
  const int lds = 2 * (32 * TM * 4 + 32 * TN * 4);
  (void)hipFuncSetAttribute((const void *)kdma_wgrad<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  kdma_wgrad<NT><<<grid, 256, lds>>>(out, ia, ig, ra, rg, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) kdma_wgrad<NT><<<grid, 256, lds>>>(out, ia, ig, ra, rg, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  double flops = (double)grid * 4 * iters * 16 * NT * 4096.0;
  printf("%-58s NT=%d grid=%d steps/block=%d  %.1f us  %.1f TFLOP/s\n", name, NT, grid, iters, ms * 1e3, flops / ms / 1e9);
  (void)hipFree(out); (void)hipFree(ra); (void)hipFree(rg); (void)hipFree(ia); (void)hipFree(ig);
}

template <int NT>
void run_gather(const char *name, int grid, int iters, int n_rows, bool random_data, int dma_bufs = 0) {
  const int ca = iters * 32;
  float *out; (void)hipMalloc(&out, sizeof(float) * (size_t)grid * 128 * 32 * NT + 1024);
  int *idx; float *rows, *wtab;
  (void)hipMalloc(&idx, sizeof(int) * (1 << 20)); (void)hipMalloc(&rows, sizeof(float) * (size_t)n_rows * ca); (void)hipMalloc(&wtab, sizeof(float) * 27 * (size_t)ca * 32 * NT);
  std::vector<int> h(1 << 20); unsigned x = 12345u;
  for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (int)((x >> 8) % (unsigned)n_rows); }
  (void)hipMemcpy(idx, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice);
  std::vector<float> hr((size_t)n_rows * ca), hw(27 * (size_t)ca * 32 * NT);
  for (auto &v : hr) { x = x * 1664525u + 1013904223u; v = random_data ? ((x >> 9) / 8388608.0f - 1.0f) : 0.01f; }
  for (auto &v : hw) { x = x * 1664525u + 1013904223u; v = random_data ? 0.05f * ((x >> 9) / 8388608.0f - 1.0f) : 0.02f; }
  (void)hipMemcpy(rows, hr.data(), sizeof(float) * hr.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(wtab, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int lds2 = 2 * (128 * 128 + 32 * NT * 128), lds3 = 3 * (128 * 128 + 32 * NT * 128);
  (void)hipFuncSetAttribute((const void *)kdma<NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
  (void)hipFuncSetAttribute((const void *)kdma<NT, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
  auto launch = [&]() {
    if (dma_bufs == 2) kdma<NT, 2><<<grid, 256, lds2>>>(out, idx, rows, wtab, iters, ca);
    else if (dma_bufs == 3) kdma<NT, 3><<<grid, 256, lds3>>>(out, idx, rows, wtab, iters, ca);
    else kgather<NT, 0><<<grid, 256>>>(out, idx, rows, wtab, iters, ca);
  };
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  double flops = (double)grid * 4 * iters * 16 * NT * 4096.0;
  printf("%-58s NT=%d grid=%d steps/block=%d  %.1f us  %.1f TFLOP/s\n", name, NT, grid, iters, ms * 1e3, flops / ms / 1e9);
  (void)hipFree(out); (void)hipFree(idx); (void)hipFree(rows); (void)hipFree(wtab);
}

template <int NT, int EPI, int PRO>
void run_tile(const char *name, int grid, int iters) {
  float *out; (void)hipMalloc(&out, sizeof(float) * (size_t)grid * 128 * 32 * NT + 1024);
  int *idx; float *rows;
  (void)hipMalloc(&idx, sizeof(int) * (1 << 20)); (void)hipMalloc(&rows, sizeof(float) * 32 * (size_t)(1 << 20));
  (void)hipMemset(idx, 0, sizeof(int) * (1 << 20)); (void)hipMemset(rows, 0, sizeof(float) * 32 * (size_t)(1 << 20));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  ktile<NT, EPI, PRO><<<grid, 256>>>(out, idx, rows, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) ktile<NT, EPI, PRO><<<grid, 256>>>(out, idx, rows, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  double flops = (double)grid * 4 * iters * 16 * NT * 4096.0;
  printf("%-58s NT=%d grid=%d steps/block=%d  %.1f us  %.1f TFLOP/s\n", name, NT, grid, iters, ms * 1e3, flops / ms / 1e9);
  (void)hipFree(out); (void)hipFree(idx); (void)hipFree(rows);
}

template <int NT, int MODE>
void run(const char *name, int blocks_per_cu) {
  float *out; (void)hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NT, MODE><<<grid, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NT, MODE><<<grid, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16 * NT * 4096.0;
  printf("%-44s NT=%d blocks/CU=%d  %.3f ms  %.1f TFLOP/s\n", name, NT, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<3, 0>("registers only", 1); run<3, 0>("registers only", 2); run<4, 0>("registers only", 1); run<1, 0>("registers only", 1); run<1, 0>("registers only", 2);
  run<3, 1>("LDS b128 both operands", 1); run<3, 1>("LDS b128 both operands", 2); run<3, 1>("LDS b128 both operands", 3); run<4, 1>("LDS b128 both operands", 2);
  run<3, 2>("LDS b128 A + b32 B", 1); run<3, 2>("LDS b128 A + b32 B", 2); run<3, 2>("LDS b128 A + b32 B", 3);
  run<3, 4>("b128 + 2 barriers per 48 MFMAs", 1); run<3, 4>("b128 + 2 barriers per 48 MFMAs", 2); run<3, 4>("b128 + 2 barriers per 48 MFMAs", 3);
  run<3, 5>("b128 + 7 ds_write_b128 per step, no barrier", 2); run<3, 5>("b128 + 7 ds_write_b128 per step, no barrier", 3);
  run<3, 3>("b128 + stores + 2 barriers (phased)", 1); run<3, 3>("b128 + stores + 2 barriers (phased)", 2); run<3, 3>("b128 + stores + 2 barriers (phased)", 3);
  run<4, 3>("b128 + stores + 2 barriers (phased)", 2);
  run_tile<3, 0, 0>("tile-like: 3017 blocks x 4 steps", 3017, 4);
  run_tile<3, 0, 0>("tile-like: 3017 blocks x 12 steps", 3017, 12);
  run_tile<3, 1, 0>("tile-like + 16-byte epilogue stores", 3017, 4);
  run_tile<3, 0, 1>("tile-like + dependent prologue loads", 3017, 4);
  run_tile<3, 1, 1>("tile-like + prologue loads + epilogue stores", 3017, 4);
  run_tile<4, 1, 1>("tile-like + prologue loads + epilogue stores", 2990, 4);
  run_tile<3, 1, 1>("persistent-like: 768 blocks x 16 steps + pro + epi", 768, 16);
  run_gather<3>("gather tile (81k rows, constant data)", 3017, 4, 81237, false);
  run_gather<3>("gather tile (81k rows, RANDOM data)", 3017, 4, 81237, true);
  run_gather<3>("gather tile (2k rows: L2-resident, random data)", 3017, 4, 2048, true);
  run_gather<4>("gather tile NT=4 (81k rows, random data)", 2990, 4, 81237, true);
  run_gather<3>("gather tile 12 steps (20k rows, random data)", 1000, 12, 20197, true);
  run_gather<3>("LDS-DMA tile, 2 buffers (81k rows, random data)", 3017, 4, 81237, true, 2);
  run_gather<3>("LDS-DMA tile, 3 buffers (81k rows, random data)", 3017, 4, 81237, true, 3);
  run_gather<4>("LDS-DMA tile NT=4, 2 buffers (81k rows, random)", 2990, 4, 81237, true, 2);
  run_gather<3>("LDS-DMA tile 12 steps, 2 buffers (20k rows, random)", 1000, 12, 20197, true, 2);
  run_wgrad<3>("wgrad-like LDS-DMA: 1536 blocks x 8 steps (128x96 tile)", 1536, 8, 81237);
  run_wgrad<3>("wgrad-like LDS-DMA: 768 blocks x 16 steps", 768, 16, 81237);
  run_wgrad<4>("wgrad-like LDS-DMA: 1536 blocks x 8 steps (128x128 tile)", 1536, 8, 81237);
  return 0;
}
