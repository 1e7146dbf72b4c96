// Adam over all parameters of the model in ONE launch (torch.optim.Adam's update rule with L2 weight decay, the optimizer
// common/solver/build.py:7-20 builds for the reference's trainers and SemanticTrainer.train_step steps once per batch).
//
// The model has ~300 parameter tensors (108 M floats).  A table in device memory describes them: parameter / first moment / second
// moment pointers and the element count (static), and per step the gradient pointer and the two bias-correction factors (the
// gradients are freshly allocated by autograd every step, so their addresses are not).  The work is cut into chunks of 16 K elements
// ahead of time (chunk -> tensor, offset: static), one block per chunk, 16-byte loads and stores: 7 streams of 4 B per element
// (read p, g, m, v; write p, m, v) = 3.0 GB per step at HBM rate instead of 14 multi-tensor launches.
#include "ftx_common.h"

using namespace ftx;

struct FtxAdamTensor {      // 48 bytes; mirrored by fusiontransformer_amd/optim.py
  float *p;
  float *m;
  float *v;
  const float *g;           // per step; NULL: no gradient this step, the tensor is skipped
  int64_t n;
  float step_size;          // lr / (1 - beta1^t)
  float inv_bc2_sqrt;       // 1 / sqrt(1 - beta2^t)
};

constexpr int ADAM_CHUNK = 16384;   // elements per block

__global__ __launch_bounds__(256) void adam_kernel(const FtxAdamTensor *__restrict__ table, const int32_t *__restrict__ chunk_tensor,
                                                   const int64_t *__restrict__ chunk_offset, float omb1, float beta2, float omb2, float eps,
                                                   float weight_decay) {
  const FtxAdamTensor t = table[chunk_tensor[blockIdx.x]];
  if (t.g == nullptr) return;
  const int64_t off = chunk_offset[blockIdx.x];
  const int64_t end = off + ADAM_CHUNK < t.n ? off + ADAM_CHUNK : t.n;
  // omb1 = 1 - beta1, omb2 = 1 - beta2 come from the host, formed in double like torch does (1.f - 0.999f is off by 1.3e-5 relative)
  auto update = [&](float &p, float g, float &m, float &v) {
    g = g + weight_decay * p;                       // L2 form: grad.add(param, alpha=weight_decay)
    m = m + omb1 * (g - m);                         // exp_avg.lerp_(grad, 1 - beta1)
    v = beta2 * v + omb2 * g * g;                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) * t.inv_bc2_sqrt + eps;
    p = p - t.step_size * (m / denom);
  };
  const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.m | (uintptr_t)t.v | (uintptr_t)t.g) & 15) == 0) && ((off & 3) == 0);
  if (vec) {
    const int64_t end4 = off + ((end - off) & ~(int64_t)3);
    for (int64_t i = off + (int64_t)threadIdx.x * 4; i < end4; i += 256 * 4) {
      float4 p = *(const float4 *)&t.p[i], m = *(const float4 *)&t.m[i], v = *(const float4 *)&t.v[i];
      const float4 g = *(const float4 *)&t.g[i];
      update(p.x, g.x, m.x, v.x);
      update(p.y, g.y, m.y, v.y);
      update(p.z, g.z, m.z, v.z);
      update(p.w, g.w, m.w, v.w);
      *(float4 *)&t.p[i] = p;
      *(float4 *)&t.m[i] = m;
      *(float4 *)&t.v[i] = v;
    }
    for (int64_t i = end4 + threadIdx.x; i < end; i += 256) update(t.p[i], t.g[i], t.m[i], t.v[i]);
  } else {
    for (int64_t i = off + threadIdx.x; i < end; i += 256) update(t.p[i], t.g[i], t.m[i], t.v[i]);
  }
}

extern "C" int32_t ftx_adam_chunk_elements(void) { return ADAM_CHUNK; }
extern "C" int32_t ftx_adam_tensor_bytes(void) { return (int32_t)sizeof(FtxAdamTensor); }

extern "C" int ftx_adam_step(const void *table, const int32_t *chunk_tensor, const int64_t *chunk_offset, int32_t n_chunks, double beta1,
                             double beta2, float eps, float weight_decay, void *stream) {
  FTX_REQUIRE(n_chunks >= 0, "ftx_adam_step: n_chunks < 0");
  if (n_chunks == 0) return FTX_OK;
  FTX_REQUIRE(table && chunk_tensor && chunk_offset, "ftx_adam_step: null pointer");
  FTX_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.f, "ftx_adam_step: bad hyper-parameter");
  adam_kernel<<<(unsigned)n_chunks, 256, 0, (hipStream_t)stream>>>((const FtxAdamTensor *)table, chunk_tensor, chunk_offset,
                                                                   (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps,
                                                                   weight_decay);
  return check_launch("ftx_adam_step");
}
