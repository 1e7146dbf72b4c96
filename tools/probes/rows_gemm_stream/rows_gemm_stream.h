// Measured and NOT shipped (round 3): see README.md in this directory.  Drop-in for csrc/ftx_spconv.hip above ftx_rows_gemm.
// ---------------------------------------------------------------------------------------
// Dense rows, streaming form: out[r,:] = A[r,:] @ W (+ bias) for the skinny GEMMs of the point branch (81 k rows x 32..384 -> 20..256:
// nn.Linear on point rows, heads, 1x1x1 convolutions).  The tile kernel above treats them as one "offset" of a pair list: every block
// stages its own copy of W, runs ONE to eight short chunks and leaves -- 2-3.5x their roofs (tools/bench_rows_gemm.py).  Here a block
// keeps its W column tile in LDS for its whole life and STREAMS row tiles through it: a wave owns 32 rows per iteration, its row
// fragments come straight from global memory (lane (row, half h) consumes channels 8t + 4h .. + 3: whole 16-byte loads, no staging),
// the next chunk's loads are in flight under the MFMAs, and the epilogue is 16-byte stores.  Same MFMA sequence per output element as
// the tile kernel (chunks ascending, then t, then s), so the two are bit-identical.
// ---------------------------------------------------------------------------------------
template <int NT, bool WT>
__global__ __launch_bounds__(256) void rows_gemm_stream_kernel(const float *__restrict__ A, int64_t n, const float *__restrict__ W,
                                                               const float *__restrict__ bias, int ca, int co, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float Ws[];   // [32 NT][ca + 4]: W^T tile, reduction index contiguous
  const int ws = ca + 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 32 * NT;
  // ---- W tile -> LDS, once
  if (WT) {   // W stored (co, ca): rows are contiguous along the reduction index
    const int per_row = ca >> 2;
    for (int e = tid; e < 32 * NT * per_row; e += 256) {
      const int nn = e / per_row, k4 = (e - nn * per_row) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + nn < co) v = *(const float4 *)&W[(int64_t)(n0 + nn) * ca + k4];
      *(float4 *)&Ws[nn * ws + k4] = v;
    }
  } else {    // W stored (ca, co): 16 bytes along co, transposed on the way in
    const int per_row = 8 * NT;   // float4 per k row of the tile
    for (int e = tid; e < ca * per_row; e += 256) {
      const int kk = e / per_row, n4 = (e - kk * per_row) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + n4 + 4 <= co) v = *(const float4 *)&W[(int64_t)kk * co + n0 + n4];
      Ws[(n4 + 0) * ws + kk] = v.x;
      Ws[(n4 + 1) * ws + kk] = v.y;
      Ws[(n4 + 2) * ws + kk] = v.z;
      Ws[(n4 + 3) * ws + kk] = v.w;
    }
  }
  __syncthreads();
  const float *wrow = Ws + l31 * ws + 4 * h;
  const int64_t row_tiles = ceil_div(n, 128);
  const int chunks = ca >> 5;
  for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
    const int64_t row = rt * 128 + wave * 32 + l31;
    const bool rv = row < n;
    const float *arow = A + (rv ? row : 0) * ca + 4 * h;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[j][g] = 0.f;
    float4 a[4], an[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *(const float4 *)(arow + 8 * t);
    for (int c = 0; c < chunks; ++c) {
      if (c + 1 < chunks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) an[t] = *(const float4 *)(arow + 32 * (c + 1) + 8 * t);   // next chunk in flight under the MFMAs
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float av[4] = {a[t].x, a[t].y, a[t].z, a[t].w};
        float wv[NT][4];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 w4 = *(const float4 *)(wrow + j * 32 * ws + 32 * c + 8 * t);
          wv[j][0] = w4.x; wv[j][1] = w4.y; wv[j][2] = w4.z; wv[j][3] = w4.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j][s], av[s], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = an[t];
    }
    if (rv) {   // lane (row l31, half h) holds channels 32j + 8q + 4h .. + 3 in registers 4q .. 4q + 3
      float *dst = out + row * co;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = n0 + j * 32 + 8 * q + 4 * h;
          if (col < co) {
            float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
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

// column tiles of 32 per block for the streaming kernel: as wide as 64 KB of LDS holds W^T ((ca + 4) floats per column), at most 4
static int stream_nt(int ca, int co) {
  int nt = (int)((64 * 1024) / (sizeof(float) * (size_t)(ca + 4)) / 32);
  if (nt > 4) nt = 4;
  const int need = (co + 31) / 32;
  if (nt > need) nt = need;
  return nt;
}

