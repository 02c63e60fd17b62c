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


// sums of squares of pieces of g: table row = {first element, elements, first partial}; block b belongs to the piece whose
// first partial is the last one <= b
__global__ __launch_bounds__(SQ_BLOCK) void sumsq_ranges_kernel(const float* __restrict__ x, const int64_t* __restrict__ table,
                                                                int n_ranges, float* __restrict__ partial) {
  __shared__ float red[SQ_BLOCK / 64];
  const int64_t bid = blockIdx.x;
  int lo = 0, hi = n_ranges - 1;
  const int64_t p0 = table[2];
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 3 + 2] - p0 <= bid) lo = mid; else hi = mid - 1;
  }
  const int64_t first = table[lo * 3], n = table[lo * 3 + 1], pb = table[lo * 3 + 2];
  const int64_t base = first + (bid - (pb - p0)) * SQ_PER_BLOCK, end = min(first + n, base + (int64_t)SQ_PER_BLOCK);
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
  if (threadIdx.x == 0) partial[p0 + bid] = s;
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

struct AdamConsts {
  float lr, b1, b2, eps, wd, bc2_sqrt, step_size, gs;
};

// one definition of the arithmetic for both kernels, with contraction into fused multiply-adds OFF: the compiler otherwise
// picks different fma pairings in different surroundings and the two kernels disagree in the last place
__device__ __forceinline__ void adam4(float (&pa)[4], const float4 gv, float (&ma)[4], float (&va)[4], const AdamConsts& k) {
#pragma clang fp contract(off)
  const float ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float gk = ga[i] * k.gs;
    pa[i] *= 1.f - k.lr * k.wd;
    ma[i] = k.b1 * ma[i] + (1.f - k.b1) * gk;
    va[i] = k.b2 * va[i] + (1.f - k.b2) * gk * gk;
    const float denom = sqrtf(va[i]) / k.bc2_sqrt + k.eps;
    pa[i] -= k.step_size * (ma[i] / denom);
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
  AdamConsts k;
  k.lr = lr, k.b1 = b1, k.b2 = b2, k.eps = eps, k.wd = wd, k.bc2_sqrt = bc2_sqrt, k.step_size = lr / bc1, k.gs = gs;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    const float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
    adam4(pa, gv, ma, va, k);
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

// ---------------------------------------------------------------- one pass for AdamW AND everything derived from the new weights
// The update used to be followed by three passes over the weights it had just written: the bf16 shadow -> transposed bf16
// shadow (js2t_transpose_groups: 186 MB in, 186 MB out), the fp32 masters -> gamma-scaled, row-centred LayerNorm-fold weights
// (js2t_fold_ln_weights: 184 MB in, 92 MB out).  Here the block that updates a piece of a weight matrix also writes its
// transposed bf16 image (through LDS) and, for a matrix a LayerNorm is folded into, the fold's weights and bias, while the new
// values are in registers.
// items: int64[n, 8] = {kind, off, rows, cols, first unit, fold row | -1, keep gradient (no clear), 0}
//   kind 0: elements [off, off + rows) of the flat buffers, units of ADAM_FLAT elements;
//   kind 1: a row-major [rows, cols] matrix at off (cols % 4 == 0, off % 4 == 0), units of ADAM_ROWS rows x ADAM_COLS columns; its
//           transposed image goes to lp_t + off as [cols, rows] when lp_t is given; fold (only for cols <= ADAM_COLS) names a
//           row of `folds` = js2t_fold_ln_weights's table {W, gamma, beta, bias | 0, Wf, bias_f, N, K} whose W is this matrix.
// A fold reads the NEW gamma / beta / bias: the caller updates the 1-D parameters in a launch of their own first.
constexpr int ADAM_FLAT = 8192, ADAM_ROWS = 64, ADAM_COLS = 512, ADAM_PITCH = ADAM_COLS + 2;
constexpr int ADAM_LDS = ADAM_ROWS * ADAM_PITCH * 2;  // bf16 image of a strip; 257-dword rows: the column reads below are conflict-free

__global__ __launch_bounds__(256) void adamw_items_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, uint16_t* __restrict__ lp, uint16_t* __restrict__ lp_t,
                                                          const int64_t* __restrict__ items, int n_items,
                                                          const int64_t* __restrict__ folds, float lr, float b1, float b2, float eps,
                                                          float wd, float bc1, float bc2_sqrt, const float* __restrict__ gscale_dev,
                                                          float gscale, int zero_grad, const float* __restrict__ lr_dev,
                                                          const int64_t* __restrict__ step_dev) {
  extern __shared__ __attribute__((aligned(16))) uint16_t tile[];
  AdamConsts k;
  k.gs = gscale * (gscale_dev ? *gscale_dev : 1.f);
  if (lr_dev) lr = *lr_dev;
  if (step_dev) {
    const double t = (double)(*step_dev);
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  k.lr = lr, k.b1 = b1, k.b2 = b2, k.eps = eps, k.wd = wd, k.bc2_sqrt = bc2_sqrt, k.step_size = lr / bc1;
  const int64_t bid = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {  // last item whose first unit <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid * 8 + 4] <= bid) lo = mid; else hi = mid - 1;
  }
  const int64_t* it = items + 8 * lo;
  const int64_t kind = it[0], off = it[1], R = it[2], Cc = it[3], u = bid - it[4], fold = it[5];
  if (it[6]) zero_grad = 0;  // its producer overwrites this piece of the gradient
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kind == 0) {
    const int64_t base = off + u * ADAM_FLAT, end = min(off + R, base + (int64_t)ADAM_FLAT);
    for (int64_t i = base + threadIdx.x * 4; i < end; i += 1024) {  // ranges are multiples of 4 elements (checked by the host)
      const float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
      float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
      adam4(pa, gv, ma, va, k);
      *(float4*)(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
      *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
      *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
      if (zero_grad) *(float4*)(g + i) = zero4;
      if (lp) *(uint2*)(lp + i) = make_uint2(pack_bf16x2(pa[0], pa[1]), pack_bf16x2(pa[2], pa[3]));
    }
    return;
  }
  const int64_t chunks = (Cc + ADAM_COLS - 1) / ADAM_COLS;
  const int64_t strip = u / chunks, r0 = strip * ADAM_ROWS, c0 = (u - strip * chunks) * ADAM_COLS;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t* fe = (fold >= 0 && chunks == 1) ? folds + 8 * fold : nullptr;
  float4 gam[2] = {zero4, zero4}, bet[2] = {zero4, zero4};
  if (fe) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t c = 4 * lane + 256 * j;
      if (c < Cc) gam[j] = *(const float4*)((const float*)fe[1] + c), bet[j] = *(const float4*)((const float*)fe[2] + c);
    }
  }
#pragma unroll 2
  for (int rr = 0; rr < ADAM_ROWS / 4; ++rr) {
    const int lr_ = wave * (ADAM_ROWS / 4) + rr;  // row inside the strip
    const int64_t r = r0 + lr_;
    if (r >= R) break;  // (uniform over the wave)
    float pn[2][4];
    float cs = 0.f, bs = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t cl = 4 * lane + 256 * j, c = c0 + cl;
      if (c < Cc) {
        const int64_t i = off + r * Cc + c;
        const float4 pv = *(float4*)(p + i), gv = *(float4*)(g + i), mv = *(float4*)(m + i), vv = *(float4*)(v + i);
        float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
        adam4(pa, gv, ma, va, k);
        *(float4*)(p + i) = make_float4(pa[0], pa[1], pa[2], pa[3]);
        *(float4*)(m + i) = make_float4(ma[0], ma[1], ma[2], ma[3]);
        *(float4*)(v + i) = make_float4(va[0], va[1], va[2], va[3]);
        if (zero_grad) *(float4*)(g + i) = zero4;
        const uint2 pk = make_uint2(pack_bf16x2(pa[0], pa[1]), pack_bf16x2(pa[2], pa[3]));
        if (lp) *(uint2*)(lp + i) = pk;
        *(uint32_t*)(tile + lr_ * ADAM_PITCH + cl) = pk.x;  // (rows are 4-byte aligned only)
        *(uint32_t*)(tile + lr_ * ADAM_PITCH + cl + 2) = pk.y;
        if (fe) {  // same order of additions as fold_ln_weights_kernel
          cs += (pa[0] * gam[j].x + pa[1] * gam[j].y) + (pa[2] * gam[j].z + pa[3] * gam[j].w);
          bs += (pa[0] * bet[j].x + pa[1] * bet[j].y) + (pa[2] * bet[j].z + pa[3] * bet[j].w);
#pragma unroll
          for (int e = 0; e < 4; ++e) pn[j][e] = pa[e];
        }
      }
    }
    if (fe) {
      cs = wave_sum(cs) / (float)Cc, bs = wave_sum(bs);
      uint16_t* Wf = (uint16_t*)fe[4];
      float carry = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t c = 4 * lane + 256 * j;
        if (c < Cc) *(uint2*)(Wf + r * Cc + c) = ln_fold_round4(pn[j], gam[j], cs, carry);
      }
      if (lane == 0) {
        const float* bias = (const float*)fe[3];
        ((float*)fe[5])[r] = bs + (bias ? bias[r] : 0.f);
      }
    }
  }
  if (lp_t == nullptr) return;
  __syncthreads();
  // the strip's columns out as rows of the transposed image: 8 rows of one column per thread (16 bytes)
  const int64_t ncol = min((int64_t)ADAM_COLS, Cc - c0);
  const bool vec = ((R | off) & 7) == 0;
  constexpr int LPR = ADAM_ROWS / 8;  // lanes per column: each writes 8 rows (16 bytes)
  for (int64_t cl = threadIdx.x / LPR; cl < ncol; cl += 256 / LPR) {
    const int r8 = (threadIdx.x % LPR) * 8;
    if (r0 + r8 >= R) continue;
    uint16_t e[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) e[q] = tile[(r8 + q) * ADAM_PITCH + cl];
    uint16_t* dst = lp_t + off + (c0 + cl) * R + r0 + r8;
    if (vec) {
      *(uint4*)dst = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16),
                                e[6] | ((uint32_t)e[7] << 16));
    } else {
      for (int q = 0; q < 8 && r0 + r8 + q < R; ++q) dst[q] = e[q];
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

extern "C" int js2t_sumsq_ranges(const float* g, const int64_t* table, int32_t n_ranges, int64_t n_blocks, float* partial,
                                 js2t_stream stream) {
  JS2T_CHECK(g && table && partial && n_ranges > 0 && n_blocks > 0 && n_blocks < 0x7fffffff, "sumsq_ranges: bad arguments");
  JS2T_CHECK((((uintptr_t)g) & 15) == 0, "sumsq_ranges: buffer must be 16-byte aligned");
  hipLaunchKernelGGL(sumsq_ranges_kernel, dim3((unsigned)n_blocks), dim3(SQ_BLOCK), 0, (hipStream_t)stream, g, table, (int)n_ranges, partial);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_norm_clip(const float* partial, int64_t n_partial, float max_norm, float* out2, js2t_stream stream) {
  JS2T_CHECK(partial && out2 && n_partial > 0, "norm_clip: bad arguments");
  hipLaunchKernelGGL(norm_clip_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, n_partial, max_norm, out2);
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

extern "C" int js2t_adamw_items(float* p, float* g, float* exp_avg, float* exp_avg_sq, void* lp_bf16, void* lp_t_bf16,
                                const int64_t* items, int32_t n_items, int64_t n_units, const int64_t* folds, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int64_t step, const float* gscale_dev, float gscale,
                                int zero_grad, const float* lr_dev, const int64_t* step_dev, js2t_stream stream) {
  JS2T_CHECK(p && g && exp_avg && exp_avg_sq && items && n_items > 0 && n_units > 0 && n_units < 0x7fffffff && step >= 1,
             "adamw_items: bad arguments");
  JS2T_CHECK(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq) | ((uintptr_t)lp_bf16) |
               ((uintptr_t)lp_t_bf16)) & 15) == 0, "adamw_items: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)adamw_items_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ADAM_LDS);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(adamw_items_kernel, dim3((unsigned)n_units), dim3(256), ADAM_LDS, (hipStream_t)stream, p, g, exp_avg, exp_avg_sq,
                     (uint16_t*)lp_bf16, (uint16_t*)lp_t_bf16, items, (int)n_items, folds, lr, beta1, beta2, eps, weight_decay,
                     (float)bc1, (float)sqrt(bc2), gscale_dev, gscale, zero_grad, lr_dev, step_dev);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" void js2t_adamw_items_geometry(int32_t* flat_unit, int32_t* rows, int32_t* cols) {
  *flat_unit = ADAM_FLAT, *rows = ADAM_ROWS, *cols = ADAM_COLS;
}
