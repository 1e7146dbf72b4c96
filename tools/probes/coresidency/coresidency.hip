// Does an MFMA-bound kernel that holds ONE workgroup per CU leave the CU usable for another stream's memory-bound kernels?
// A: fp32-MFMA loop with LDS fragment reads, `bpc` blocks of 256 threads per CU (long-running, like a persistent GEMM)
// B: a chain of streaming kernels (3 reads + 1 write of 32 MB each, like BatchNorm / reduce passes) on a second stream
// Printed: A alone, B alone, both (wall, and each one's own span) for bpc = 1, 2, 4 and two register budgets of A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NT>
__global__ __launch_bounds__(256) void mfma_kernel(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float As[128 * 36];
  __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < 128 * 36; i += 256) { As[i] = (float)(i % 7) * 0.01f; Bs[i] = (float)(i % 5) * 0.02f; }
  __syncthreads();
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
  const float *ap = &As[(wave * 32 + l31) * 36 + 4 * half];
  const float *bp = &Bs[l31 * 36 + 4 * half];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 a = *(const float4 *)(ap + 8 * t);
      float av[4] = {a.x, a.y, a.z, a.w};
      float bv[NT][4];
#pragma unroll
      for (int j = 0; j < NT; ++j) { float4 b = *(const float4 *)(bp + (j % 4) * 32 * 36 + 8 * t); bv[j][0] = b.x; bv[j][1] = b.y; bv[j][2] = b.z; bv[j][3] = b.w; }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][s], av[s], acc[j], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  float s = 0;
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) s += acc[j][g];
  out[blockIdx.x * 256 + tid] = s;
}

template <int PRIO>
__global__ void stream_kernel(const float4 *a, const float4 *b, const float4 *c, float4 *o, size_t n) {
  if (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);   // raise this wave's issue priority over co-resident waves (the MFMA kernel's)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 x = a[i], y = b[i], z = c[i];
    o[i] = make_float4(x.x + y.x * z.x, x.y + y.y * z.y, x.z + y.z * z.z, x.w + y.w * z.w);
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT, int PRIO>
int run(int bpc, int iters) {
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  const size_t n4 = (32u << 20) / 16;   // 32 MB per array
  float4 *a, *b, *c, *o;
  CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&c, n4 * 16)); CK(hipMalloc(&o, n4 * 16));
  CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16)); CK(hipMemset(c, 0, n4 * 16));
  hipEvent_t ev[6];
  for (auto &x : ev) CK(hipEventCreate(&x));
  const int chain = 60;
  auto A = [&]() { mfma_kernel<NT><<<256 * bpc, 256, 0, sa>>>(out, iters); };
  auto B = [&]() { for (int i = 0; i < chain; ++i) stream_kernel<PRIO><<<2048, 256, 0, sb>>>(a, b, c, o, n4); };
  A(); B(); CK(hipDeviceSynchronize());
  float ta, tb, tab_a, tab_b;
  CK(hipEventRecord(ev[0], sa)); A(); CK(hipEventRecord(ev[1], sa)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ta, ev[0], ev[1]));
  CK(hipEventRecord(ev[2], sb)); B(); CK(hipEventRecord(ev[3], sb)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&tb, ev[2], ev[3]));
  CK(hipEventRecord(ev[0], sa)); CK(hipEventRecord(ev[2], sb)); A(); B(); CK(hipEventRecord(ev[1], sa)); CK(hipEventRecord(ev[3], sb));
  CK(hipDeviceSynchronize());
  CK(hipEventElapsedTime(&tab_a, ev[0], ev[1])); CK(hipEventElapsedTime(&tab_b, ev[2], ev[3]));
  const double flops = 2.0 * 32 * 32 * 2 * 16 * NT * 4.0 * iters * 256 * bpc;
  printf("B at s_setprio %d | A: %d accumulators, %d block(s)/CU: A alone %.2f ms (%.0f TFLOP/s) | B alone %.2f ms (%.0f GB/s) | together: A %.2f ms, B %.2f ms  (sum alone %.2f, max alone %.2f)\n",
         PRIO, NT, bpc, ta, flops / ta / 1e9, tb, chain * 4.0 * 32 * (1 << 20) / tb / 1e6, tab_a, tab_b, ta + tb, ta > tb ? ta : tb);
  return 0;
}

int main() {
  if (run<4, 0>(1, 6000)) return 1;
  if (run<4, 0>(2, 3000)) return 1;
  if (run<4, 0>(4, 1500)) return 1;
  if (run<8, 0>(1, 3000)) return 1;
  if (run<8, 0>(2, 1500)) return 1;
  if (run<4, 3>(1, 6000)) return 1;
  if (run<4, 3>(2, 3000)) return 1;
  if (run<4, 3>(4, 1500)) return 1;
  if (run<8, 3>(2, 1500)) return 1;
  return 0;
}
