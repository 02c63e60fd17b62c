// Update tail of the train step over the flat parameter store: global gradient norm (clip_grad_norm_,
// builders.py:68-71) and fused AdamW (builders.py:112-114) that also re-casts the bf16 shadow and clears the
// gradient buffer — one pass over 4 fp32 streams instead of torch's multi-tensor chains.
#include "common.hpp"

namespace {

constexpr int SQ_BLOCK = 256;
constexpr int SQ_PER_BLOCK = 256 * 16 * 4;  // elements reduced by one block

__global__ __launch_bounds__(SQ_BLOCK) void sumsq_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[SQ_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * SQ_PER_BLOCK;
  const int64_t end = min(n, base + SQ_PER_BLOCK);
  float s = 0.f;
  for (int64_t i = base + threadIdx.x * 4; i < end; i += SQ_BLOCK * 4) {
    if (i + 4 <= end) {
      const float4 v = *(const float4*)(x + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (int64_t j = i; j < end; ++j) s += x[j] * x[j];
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = sqrt(sum partial) (global L2 norm), out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(1024) void norm_clip_kernel(const float* __restrict__ partial, int64_t np, float max_norm,
                                                         float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < np; i += 1024) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    out[0] = norm;
    out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  }
}

// torch.optim.AdamW (decoupled weight decay), single-tensor formula order:
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             uint16_t* __restrict__ lp, int64_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, const float* __restrict__ gscale_dev, float gscale, int zero_grad,
                             const float* __restrict__ lr_dev, const int64_t* __restrict__ step_dev) {
  const float gs = gscale * (gscale_dev ? *gscale_dev : 1.f);
  if (lr_dev) lr = *lr_dev;
  if (step_dev) {  // device-resident update count (hipGraph replays cannot change kernel arguments)
    const double t = (double)(*step_dev);
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  const float step_size = lr / bc1;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w},
          va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = ga[k] * gs;
      pa[k] *= 1.f - lr * wd;
      ma[k] = b1 * ma[k] + (1.f - b1) * gk;
      va[k] = b2 * va[k] + (1.f - b2) * gk * gk;
      const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
      pa[k] -= step_size * (ma[k] / denom);
    }
    *(float4*)(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
    if (zero_grad) *(float4*)(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lp) {
      uint2 pk;
      pk.x = pack_bf16x2(pa[0], pa[1]);
      pk.y = pack_bf16x2(pa[2], pa[3]);
      *(uint2*)(lp + i) = pk;
    }
  }
}

}  // namespace

extern "C" int64_t js2t_sumsq_partials(int64_t n) { return (n + SQ_PER_BLOCK - 1) / SQ_PER_BLOCK; }

extern "C" int js2t_grad_norm_clip(const float* g, int64_t n, float max_norm, float* partial, float* out2, js2t_stream stream) {
  JS2T_CHECK(g && partial && out2 && n > 0, "grad_norm_clip: bad arguments");
  JS2T_CHECK((((uintptr_t)g) & 15) == 0, "grad_norm_clip: buffer must be 16-byte aligned");
  const int64_t np = js2t_sumsq_partials(n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)np), dim3(SQ_BLOCK), 0, s, g, n, partial);
  JS2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_clip_kernel, dim3(1), dim3(1024), 0, s, partial, np, max_norm, out2);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_adamw(float* p, float* g, float* exp_avg, float* exp_avg_sq, void* lp_bf16, int64_t n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int64_t step, const float* gscale_dev, float gscale,
                          int zero_grad, const float* lr_dev, const int64_t* step_dev, js2t_stream stream) {
  JS2T_CHECK(p && g && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw: bad arguments");
  JS2T_CHECK(n % 4 == 0, "adamw: flat buffers must be padded to a multiple of 4 elements");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int64_t grid = (n / 4 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq,
                     (uint16_t*)lp_bf16, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), gscale_dev, gscale,
                     zero_grad, lr_dev, step_dev);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
