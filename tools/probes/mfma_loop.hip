// Inner-loop ceilings of the fp32 MFMA on gfx950: registers only, with LDS fragment reads (b128 / b32), 1-2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
    } else if (MODE == 1) {  // b128 for both operands
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
  }
  float s = 0.f;
  for (int j = 0; j < NT; ++j) for (int g = 0; g < 16; ++g) s += acc[j][g];
  out[blockIdx.x * 256 + tid] = s;
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
  return 0;
}
