// Row-wise HBM-bound kernels with wavefront-shuffle reductions: LayerNorm fwd/bwd and the masked
// attention softmax (+ dropout) fwd/bwd, plus the head-mean of attention weights.
// One 64-lane wave owns one row; a 256-thread block holds 4 rows.
#include "common.hpp"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves

// ---------------------------------------------------------------- LayerNorm forward
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int64_t rows, int64_t D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * D;
  float s = 0.f;
  for (int64_t c = lane; c < D; c += 64) s += io<T>::ld(xr + c);
  const float mu = wave_sum(s) / (float)D;
  float v = 0.f;
  for (int64_t c = lane; c < D; c += 64) {
    const float d = io<T>::ld(xr + c) - mu;
    v += d * d;
  }
  const float rs = rsqrtf(wave_sum(v) / (float)D + eps);
  T* yr = y + row * D;
  for (int64_t c = lane; c < D; c += 64) io<T>::st(yr + c, (io<T>::ld(xr + c) - mu) * rs * gamma[c] + beta[c]);
  if (lane == 0) {
    mean[row] = mu;
    rstd[row] = rs;
  }
}

// ---------------------------------------------------------------- LayerNorm backward (dx) + partial dgamma/dbeta
// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, T* __restrict__ dx,
                                                               const T* __restrict__ add, float add_scale,
                                                               int64_t rows, int64_t D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float mu = mean[row], rs = rstd[row];
  const T* xr = x + row * D;
  const T* gr = dy + row * D;
  float s1 = 0.f, s2 = 0.f;
  for (int64_t c = lane; c < D; c += 64) {
    const float g = io<T>::ld(gr + c) * gamma[c];
    const float xh = (io<T>::ld(xr + c) - mu) * rs;
    s1 += g;
    s2 += g * xh;
  }
  s1 = wave_sum(s1) / (float)D;
  s2 = wave_sum(s2) / (float)D;
  T* dr = dx + row * D;
  for (int64_t c = lane; c < D; c += 64) {
    const float g = io<T>::ld(gr + c) * gamma[c];
    const float xh = (io<T>::ld(xr + c) - mu) * rs;
    float v = rs * (g - s1 - xh * s2);
    if (add) v += add_scale * io<T>::ld(add + row * D + c);
    io<T>::st(dr + c, v);
  }
}
// partial[(2*part + 0)*D + c] = sum_r dy*xhat ; partial[(2*part+1)*D + c] = sum_r dy   over a 128-row slab
// (block = 64 columns x 4 row-lanes, like colsum_partial_kernel)
constexpr int LN_SLAB = 256;
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_param_partial_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                          const float* __restrict__ mean,
                                                                          const float* __restrict__ rstd,
                                                                          float* __restrict__ partial, int64_t rows, int64_t D) {
  __shared__ float shg[4][64], shb[4][64];
  const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * LN_SLAB, r1 = min(rows, r0 + LN_SLAB);
  float sg = 0.f, sb = 0.f;
  if (c < D)
    for (int64_t r = r0 + sub; r < r1; r += 4) {
      const float g = io<T>::ld(dy + r * D + c);
      sg += g * (io<T>::ld(x + r * D + c) - mean[r]) * rstd[r];
      sb += g;
    }
  shg[sub][cl] = sg;
  shb[sub][cl] = sb;
  __syncthreads();
  if (sub == 0 && c < D) {
    partial[((int64_t)blockIdx.y * 2 + 0) * D + c] = (shg[0][cl] + shg[1][cl]) + (shg[2][cl] + shg[3][cl]);
    partial[((int64_t)blockIdx.y * 2 + 1) * D + c] = (shb[0][cl] + shb[1][cl]) + (shb[2][cl] + shb[3][cl]);
  }
}
__global__ __launch_bounds__(256) void layernorm_bwd_param_final_kernel(const float* __restrict__ partial,
                                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                        int64_t nparts, int64_t D, int accumulate) {
  // block = 64 columns x 4 partial-lanes; fixed summation order -> bit-reproducible
  __shared__ float shg[4][64], shb[4][64];
  const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  float sg0 = 0.f, sg1 = 0.f, sb0 = 0.f, sb1 = 0.f;
  if (c < D) {
    int64_t p = sub;
    for (; p + 4 < nparts; p += 8) {
      sg0 += partial[(p * 2 + 0) * D + c];
      sb0 += partial[(p * 2 + 1) * D + c];
      sg1 += partial[((p + 4) * 2 + 0) * D + c];
      sb1 += partial[((p + 4) * 2 + 1) * D + c];
    }
    for (; p < nparts; p += 4) {
      sg0 += partial[(p * 2 + 0) * D + c];
      sb0 += partial[(p * 2 + 1) * D + c];
    }
  }
  shg[sub][cl] = sg0 + sg1;
  shb[sub][cl] = sb0 + sb1;
  __syncthreads();
  if (sub == 0 && c < D) {
    const float tg = (shg[0][cl] + shg[1][cl]) + (shg[2][cl] + shg[3][cl]);
    const float tb = (shb[0][cl] + shb[1][cl]) + (shb[2][cl] + shb[3][cl]);
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + tg;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + tb;
  }
}

// ---------------------------------------------------------------- vectorised LayerNorm (D % 8 == 0, D <= 2048)
// One wave per row, every lane owns 8-column chunks (16-byte bf16 / 2 x 16-byte f32 accesses).  The backward kernel
// also accumulates this block's dgamma/dbeta partial sums in registers while it streams dy and x for dx, so the
// parameter gradients cost no extra pass over the activations.
template <typename T> struct ld8;
template <> struct ld8<uint16_t> {
  static __device__ __forceinline__ void ld(const uint16_t* p, float (&v)[8]) {
    const uint4 r = *(const uint4*)p;
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(rr[i] << 16); v[2 * i + 1] = __uint_as_float(rr[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(uint16_t* p, const float (&v)[8]) {
    uint4 r;
    r.x = pack_bf16x2(v[0], v[1]);
    r.y = pack_bf16x2(v[2], v[3]);
    r.z = pack_bf16x2(v[4], v[5]);
    r.w = pack_bf16x2(v[6], v[7]);
    *(uint4*)p = r;
  }
};
template <> struct ld8<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
constexpr int LNV_MAXCH = 4;        // chunks of 8 columns per lane -> D <= 2048
constexpr int LNV_ROWS = 64;        // most rows per block in the backward kernel (shared by its 4..16 waves)
constexpr int LNV_MIN_ROWS = 16;    // fewest (sizes the caller's partial-sum workspace: 2 * ceil(rows / 16) * D floats)

// y8 / q_state (optional, js2t_layernorm_fwd_fp8): the result ALSO (y may then be NULL: only) as e4m3 bytes with a delayed
// per-tensor scale - q_state[0] = scale in use, [1] = running max |y| of this call.  The quantisation of a LayerNorm-fed
// nn.Linear input then costs no pass of its own: the normalised row is in registers here anyway.
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, T* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                                int D, float eps, uint8_t* __restrict__ y8 = nullptr,
                                                                float* __restrict__ q_state = nullptr, const float* __restrict__ q_mul = nullptr,
                                                                float* __restrict__ q_scale_out = nullptr) {
  const int lane = threadIdx.x & 63;
  __shared__ float q_red[ROWS_PER_BLOCK];
  float q_inv = 0.f, q_max = 0.f;
  if (y8) {
    const float S = q_state[0];
    q_inv = S > 0.f ? 1.f / S : 0.f;
    if (q_scale_out && blockIdx.x == 0 && threadIdx.x == 0) *q_scale_out = (S > 0.f ? S : 1.f) * (q_mul ? *q_mul : 1.f);
  }
  for (int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * ROWS_PER_BLOCK) {
  const int nch = D >> 3;
  float v[LNV_MAXCH][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LNV_MAXCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      ld8<T>::ld(x + row * D + 8 * c, v[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[j][i];
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LNV_MAXCH; ++j)
    if (lane + 64 * j < nch) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float dlt = v[j][i] - mu; q += dlt * dlt; }
    }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < LNV_MAXCH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      float gm[8], bt[8], o[8];
      ld8<float>::ld(gamma + 8 * c, gm);
      ld8<float>::ld(beta + 8 * c, bt);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (v[j][i] - mu) * rs * gm[i] + bt[i];
      if (y) ld8<T>::st(y + row * D + 8 * c, o);
      if (y8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          q_max = fmaxf(q_max, fabsf(o[i]));
          o[i] = fminf(fmaxf(o[i] * q_inv, -448.f), 448.f);
        }
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0], o[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2], o[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4], o[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6], o[7], hi, true);
        *(uint2*)(y8 + row * D + 8 * c) = make_uint2((uint32_t)lo, (uint32_t)hi);
      }
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }  // rows of this wave
  if (y8) {
    // delayed scaling: post this block's max |y| - only if it beats what is there, so a handful of blocks do.  The hand-over
    // (q_state[0] = q_state[1] / 448, q_state[1] = 0) is left to the consuming product (js2t_gemm fp8_state): it runs between
    // this call and the next one on the stream, so no arrival ticket - 3000 same-address atomics, ~0.1 ms - is needed here.
    q_max = wave_max(q_max);
    if (lane == 0) q_red[threadIdx.x >> 6] = q_max;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = 0.f;
      for (int i = 0; i < ROWS_PER_BLOCK; ++i) m = fmaxf(m, q_red[i]);
      unsigned int* st = (unsigned int*)q_state;
      if (__float_as_uint(m) > *(volatile unsigned int*)(st + 1)) atomicMax(st + 1, __float_as_uint(m));
    }
  }
}

// dx (+ add) and per-block partial sums of dgamma / dbeta: partial[(2*blk + 0)*D + c], partial[(2*blk + 1)*D + c].
// NCH = chunks of 8 columns per lane (compile-time so the per-lane arrays are no larger than the row needs); the block
// has blockDim/64 waves sharing LNV_ROWS rows - many short waves, because one row is a load -> two wave reductions ->
// store latency chain and only other waves can hide it.
template <typename T, int NCH>
__global__ __launch_bounds__(1024) void layernorm_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, T* __restrict__ dx,
                                                                 const T* __restrict__ add, float add_scale,
                                                                 float* __restrict__ partial, float* dgamma_acc,
                                                                 float* dbeta_acc, int64_t rows, int D, T* __restrict__ dxd,
                                                                 float drop_p, const uint64_t* __restrict__ rng, uint32_t rng_stream,
                                                                 int rows_per_block, int acc_copies, int64_t acc_stride,
                                                                 const float* __restrict__ beta, T* __restrict__ n_out) {
  extern __shared__ float red[];  // [waves][2][D]
  // optional third output n_out = xhat * gamma + beta, the LayerNorm's FORWARD result: with the normalisation folded into
  // the consuming product (js2t_gemm ln_stats) nobody wrote it, and the deferred weight gradient dW = dY^T n still wants it -
  // re-materialised here while x, mean and rstd are in registers
  // optional second output dxd = dropout_bwd(dx) for the mask of call site rng_stream: the block that produced this
  // LayerNorm's input starts its backward with exactly that product (one read of dx and one launch less)
  const uint32_t dkey = dxd ? dropout_key(rng, rng_stream) : 0u;
  const uint32_t dthr = (uint32_t)(drop_p * 65536.0f);
  const float dsc = 1.f / (1.f - drop_p);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nch = D >> 3;
  const int rpw = rows_per_block / nw;
  float gm[NCH][8], ag[NCH][8], ab[NCH][8], bt[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
#pragma unroll
    for (int i = 0; i < 8; ++i) { ag[j][i] = 0.f; ab[j][i] = 0.f; gm[j][i] = 0.f; bt[j][i] = 0.f; }
    if (c < nch) ld8<float>::ld(gamma + 8 * c, gm[j]);
    if (n_out && c < nch) ld8<float>::ld(beta + 8 * c, bt[j]);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block + w * rpw;
  for (int rr = 0; rr < rpw; ++rr) {
    const int64_t row = r0 + rr;
    if (row >= rows) break;
    const float mu = mean[row], rs = rstd[row];
    float g[NCH][8], xh[NCH][8], av[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float dyv[8], xv[8];
        ld8<T>::ld(dy + row * D + 8 * c, dyv);
        ld8<T>::ld(x + row * D + 8 * c, xv);
        if (add) ld8<T>::ld(add + row * D + 8 * c, av[j]);  // requested with the other two rows: off the reduction's critical path
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          xh[j][i] = (xv[i] - mu) * rs;
          ag[j][i] += dyv[i] * xh[j][i];
          ab[j][i] += dyv[i];
          g[j][i] = dyv[i] * gm[j][i];
          s1 += g[j][i];
          s2 += g[j][i] * xh[j][i];
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rs * (g[j][i] - s1 - xh[j][i] * s2);
        if (add) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += add_scale * av[j][i];
        }
        ld8<T>::st(dx + row * D + 8 * c, o);
        if (n_out) {
          float nv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) nv[i] = fmaf(xh[j][i], gm[j][i], bt[j][i]);
          ld8<T>::st(n_out + row * D + 8 * c, nv);
        }
        if (dxd) {  // the decisions of dropout_keep4_key(dkey, row, 2c) and (.., 2c + 1), straight from the hash halves
          float od[8];
          const uint32_t rowkey = hash32((uint32_t)row ^ dkey) + 4u * (uint32_t)c;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t h = hash32w(rowkey + (uint32_t)q);
            od[2 * q] = (h & 0xffffu) >= dthr ? o[2 * q] * dsc : 0.f;
            od[2 * q + 1] = (h >> 16) >= dthr ? o[2 * q + 1] * dsc : 0.f;
          }
          ld8<T>::st(dxd + row * D + 8 * c, od);
        }
      }
    }
  }
  if (partial || dgamma_acc) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c < nch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          red[(w * 2 + 0) * D + 8 * c + i] = ag[j][i];
          red[(w * 2 + 1) * D + 8 * c + i] = ab[j][i];
        }
      }
    }
    __syncthreads();
    typedef __attribute__((address_space(1))) float gfloat;
    for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) {
      const int which = i / D, c = i - which * D;
      float tot = 0.f;
      for (int ww = 0; ww < nw; ++ww) tot += red[(ww * 2 + which) * D + c];
      if (dgamma_acc) {
        // accumulate mode: the block totals go straight onto the gradient (contiguous 1 KB atomic segments per
        // wave-instruction), no partial slab and no second kernel
        // every block adds 2 D values onto the same 2 D addresses: a few hundred same-address atomics in a row cost more
        // than the whole dx pass (12000 x 512: 21.7 us against 9.95 without them).  With acc_copies > 1 block b adds into
        // copy b % acc_copies of a workspace (12.8 us with 8 copies); js2t_fold_copies sums the copies once per step.
        __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)(which ? dbeta_acc : dgamma_acc) + (blockIdx.x % acc_copies) * acc_stride + c, tot);
      } else {
        partial[((int64_t)blockIdx.x * 2 + which) * D + c] = tot;
      }
    }
  }
}

// ---------------------------------------------------------------- masked softmax (+dropout) forward
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* S, const uint8_t* __restrict__ mask,
                                                          T* P, T* Pd, int64_t B, int64_t H,
                                                          int64_t Tq, int64_t Tk, int64_t ld, int64_t mask_sb,
                                                          int64_t mask_sq, float p, const uint64_t* rng, uint32_t stream) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);  // over Z*Tq
  if (row >= B * H * Tq) return;
  const int64_t z = row / Tq, q = row - z * Tq, b = z / H;
  const T* sr = S + row * ld;
  const uint8_t* mr = mask ? mask + b * mask_sb + q * mask_sq : nullptr;
  float mx = -INFINITY;
  for (int64_t k = lane; k < Tk; k += 64) {
    const bool on = !mr || mr[k];
    if (on) mx = fmaxf(mx, io<T>::ld(sr + k));
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int64_t k = lane; k < Tk; k += 64) {
    const bool on = !mr || mr[k];
    if (on) sum += __expf(io<T>::ld(sr + k) - mx);
  }
  sum = wave_sum(sum);
  // A fully masked row yields NaN exactly as softmax over all -inf does in the reference.
  const float inv = 1.f / sum;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t dkey = p > 0.f ? dropout_key(rng, stream) : 0u;
  T* pr = P + row * ld;
  T* pdr = Pd + row * ld;
  for (int64_t k4 = lane; k4 * 4 < ld; k4 += 64) {
    uint32_t keep = 0xFu;
    if (p > 0.f) keep = dropout_keep4_key(dkey, (uint32_t)row, (uint32_t)k4, p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = 4 * k4 + i;
      if (k >= ld) break;
      float v = 0.f;
      if (k < Tk) {
        const bool on = !mr || mr[k];
        v = on ? __expf(io<T>::ld(sr + k) - mx) * inv : (sum > 0.f ? 0.f : NAN);
      }
      io<T>::st(pr + k, v);
      if (Pd != P) io<T>::st(pdr + k, ((keep >> i) & 1u) ? v * sc : 0.f);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, const T* __restrict__ dPd,
                                                          T* __restrict__ dS, int64_t nrows, int64_t Tk, int64_t ld, float p,
                                                          const uint64_t* rng, uint32_t stream) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const T* pr = P + row * ld;
  const T* gr = dPd + row * ld;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t dkey = p > 0.f ? dropout_key(rng, stream) : 0u;
  float dot = 0.f;
  for (int64_t k4 = lane; k4 * 4 < Tk; k4 += 64) {
    uint32_t keep = 0xFu;
    if (p > 0.f) keep = dropout_keep4_key(dkey, (uint32_t)row, (uint32_t)k4, p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = 4 * k4 + i;
      if (k < Tk && ((keep >> i) & 1u)) dot += io<T>::ld(gr + k) * sc * io<T>::ld(pr + k);
    }
  }
  dot = wave_sum(dot);
  T* dr = dS + row * ld;
  for (int64_t k4 = lane; k4 * 4 < ld; k4 += 64) {
    uint32_t keep = 0xFu;
    if (p > 0.f) keep = dropout_keep4_key(dkey, (uint32_t)row, (uint32_t)k4, p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = 4 * k4 + i;
      if (k >= ld) break;
      float v = 0.f;
      if (k < Tk) {
        const float g = ((keep >> i) & 1u) ? io<T>::ld(gr + k) * sc : 0.f;
        v = io<T>::ld(pr + k) * (g - dot);
      }
      io<T>::st(dr + k, v);
    }
  }
}

template <typename T>
__global__ void attn_head_mean_kernel(const T* __restrict__ P, float* __restrict__ out, int64_t B, int64_t H, int64_t Tq,
                                      int64_t Tk, int64_t ld) {
  const int64_t total = B * Tq * Tk;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (Tq * Tk), rem = i - b * Tq * Tk, q = rem / Tk, k = rem - q * Tk;
    float s = 0.f;
    for (int64_t h = 0; h < H; ++h) s += io<T>::ld(P + ((b * H + h) * Tq + q) * ld + k);
    out[i] = s / (float)H;
  }
}

// Relative-position bias on the MATERIALISED attention path (extension, see js2t_rel_bias_add): the fused kernels add the
// term on chip; this pair serves fp32 compute / odd head sizes, i.e. the parity tests of the composed config-5 model.
template <typename T>
__global__ void rel_bias_add_kernel(T* __restrict__ S, const float* __restrict__ bias, int64_t B, int64_t H, int64_t Tq,
                                    int64_t Tk, int64_t ld, int R) {
  const int64_t total = B * H * Tq * Tk;
  const int W = 2 * R + 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / Tk, k = i - row * Tk, q = row % Tq, h = (row / Tq) % H;
    const int64_t dlt = k - q;
    const int r = (int)(dlt < -R ? -R : (dlt > R ? R : dlt)) + R;
    T* at = S + row * ld + k;
    io<T>::st(at, io<T>::ld(at) + bias[h * W + r]);
  }
}

// d_bias[h, r] += sum over (b, q, k with clamp(k - q) == r) of dS: one block per (head, slab of query rows), an LDS
// histogram of 2R + 1 bins per block, one global atomic per bin and block
template <typename T>
__global__ void rel_bias_grad_kernel(const T* __restrict__ dS, float* __restrict__ d_bias, int64_t B, int64_t H, int64_t Tq,
                                     int64_t Tk, int64_t ld, int R, int64_t rows_per_block) {
  extern __shared__ float hist[];
  const int W = 2 * R + 1;
  const int64_t h = blockIdx.y;
  for (int i = threadIdx.x; i < W; i += blockDim.x) hist[i] = 0.f;
  __syncthreads();
  const int64_t r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, B * Tq);  // rows of this head: (b, q) pairs
  for (int64_t i = r0 * Tk + threadIdx.x; i < r1 * Tk; i += blockDim.x) {
    const int64_t bq = i / Tk, k = i - bq * Tk, b = bq / Tq, q = bq - b * Tq;
    const float g = io<T>::ld(dS + ((b * H + h) * Tq + q) * ld + k);
    const int64_t dlt = k - q;
    const int r = (int)(dlt < -R ? -R : (dlt > R ? R : dlt)) + R;
    if (g != 0.f) atomicAdd(&hist[r], g);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W; i += blockDim.x)
    if (hist[i] != 0.f) atomicAdd(d_bias + h * W + i, hist[i]);
}

// ---------------------------------------------------------------- weights of the LayerNorm fold
// table row e: {W f32[N,K], gamma f32[K], beta f32[K], bias f32[N] | 0, Wf bf16[N,K], bias_f f32[N], N, K}; one wave per weight
// row: c = sum_k W gamma, Wf = bf16(W gamma - c / K), rounded with error feedback (a centred row: the product of a raw activation row with it equals the
// product of the mean-free row with W gamma), bias_f = bias + W . beta
__global__ __launch_bounds__(256) void fold_ln_weights_kernel(const int64_t* __restrict__ table) {
  const int64_t* e = table + 8 * blockIdx.y;
  const float* W = (const float*)e[0];
  const float* gamma = (const float*)e[1];
  const float* beta = (const float*)e[2];
  const float* bias = (const float*)e[3];
  uint16_t* Wf = (uint16_t*)e[4];
  float* bias_f = (float*)e[5];
  const int64_t N = e[6], K = e[7];
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float cs = 0.f, bs = 0.f;
  for (int64_t k = 4 * lane; k < K; k += 256) {  // K % 4 == 0 (checked by the host)
    const float4 w = *(const float4*)(W + n * K + k), g = *(const float4*)(gamma + k), b = *(const float4*)(beta + k);
    cs += (w.x * g.x + w.y * g.y) + (w.z * g.z + w.w * g.w);
    bs += (w.x * b.x + w.y * b.y) + (w.z * b.z + w.w * b.w);
  }
  cs = wave_sum(cs) / (float)K, bs = wave_sum(bs);
  float carry = 0.f;  // rounding with error feedback along the lane's own elements (ln_fold_round4)
  for (int64_t k = 4 * lane; k < K; k += 256) {
    const float4 w = *(const float4*)(W + n * K + k), g = *(const float4*)(gamma + k);
    const float wa[4] = {w.x, w.y, w.z, w.w};
    *(uint2*)(Wf + n * K + k) = ln_fold_round4(wa, g, cs, carry);
  }
  if (lane == 0) bias_f[n] = bs + (bias ? bias[n] : 0.f);
}

}  // namespace

template <typename T, int NCH>
static int launch_ln_bwd_vec(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                             const void* add, float add_scale, float* part, float* dga, float* dba, int64_t rows, int D,
                             int64_t nblk, int nw, size_t lds, hipStream_t s, void* dxd, float drop_p, const uint64_t* rng,
                             uint32_t rng_stream, int rows_per_block, int acc_copies, int64_t acc_stride, const float* beta,
                             void* n_out) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)layernorm_bwd_vec_kernel<T, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((layernorm_bwd_vec_kernel<T, NCH>), dim3((unsigned)nblk), dim3(64 * nw), lds, s, (const T*)dy, (const T*)x, gamma,
                     mean, rstd, (T*)dx, (const T*)add, add_scale, part, dga, dba, rows, D, (T*)dxd, drop_p, rng, rng_stream, rows_per_block, acc_copies, acc_stride,
                     beta, (T*)n_out);
  return JS2T_OK;
}

#define DISPATCH_DT(dt, T, ...)                                  \
  do {                                                           \
    if ((dt) == JS2T_F32) { typedef float T; __VA_ARGS__; }      \
    else if ((dt) == JS2T_BF16) { typedef uint16_t T; __VA_ARGS__; } \
    else { js2t_set_error("bad dtype %d", (int)(dt)); return JS2T_ERR_INVALID; } \
  } while (0)

extern "C" int64_t js2t_colsum_partial_rows(int64_t rows);

extern "C" int js2t_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                  int64_t rows, int64_t D, float eps, int dt, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(x && gamma && beta && y && mean && rstd && rows > 0 && D > 0, "layernorm_fwd: bad arguments");
  const bool vec = (D % 8 == 0) && D <= 64 * 8 * LNV_MAXCH &&
                   (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
  if (vec) {
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((layernorm_fwd_vec_kernel<T>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0,
                                          (hipStream_t)stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, (int)D, eps));
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, D, eps));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y, void* y8, float* q_state,
                                      const float* q_mul, float* q_scale_out, float* mean, float* rstd, int64_t rows, int64_t D, float eps,
                                      int dt, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(x && gamma && beta && y8 && q_state && mean && rstd && rows > 0 && D > 0, "layernorm_fwd_fp8: bad arguments");
  JS2T_CHECK((D % 8 == 0) && D <= 64 * 8 * LNV_MAXCH && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)q_state) & 15) == 0 &&
                 (((uintptr_t)y8) & 7) == 0,
             "layernorm_fwd_fp8: D % 8 == 0, D <= 2048, 16-byte aligned rows / state");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((layernorm_fwd_vec_kernel<T>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, (int)D, eps, (uint8_t*)y8,
                                        q_state, q_mul, q_scale_out));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                  void* dx, const void* add, float add_scale, float* dgamma, float* dbeta, float* partial,
                                  int accumulate, int64_t rows, int64_t D, int dt, js2t_stream stream) {
  return js2t_layernorm_bwd_dropout(dy, x, gamma, mean, rstd, dx, add, add_scale, dgamma, dbeta, partial, accumulate, rows, D, dt,
                                    nullptr, 0.f, nullptr, 0u, 1, 0, stream);
}

// dst_e[i] += sum over copies of ws[off_e + c * stride + i]; the copies are zeroed for the next step.  table = int64[n][3]:
// {workspace offset in floats, destination pointer, count}.  One block per entry.
__global__ __launch_bounds__(256) void fold_copies_kernel(float* __restrict__ ws, const int64_t* __restrict__ table, int copies,
                                                          int64_t stride) {
  const int64_t off = table[blockIdx.x * 3], count = table[blockIdx.x * 3 + 2];
  float* dst = (float*)table[blockIdx.x * 3 + 1];
  for (int64_t i = threadIdx.x; i < count; i += blockDim.x) {
    float sum = 0.f;
    for (int c = 0; c < copies; ++c) {
      sum += ws[off + c * stride + i];
      ws[off + c * stride + i] = 0.f;
    }
    dst[i] += sum;
  }
}
extern "C" int js2t_fold_copies(float* ws, const int64_t* table, int32_t n_entries, int32_t copies, int64_t copy_stride,
                                js2t_stream stream) {
  if (n_entries == 0) return JS2T_OK;
  JS2T_CHECK(ws && table && n_entries > 0 && copies >= 1 && copy_stride >= 0, "fold_copies: bad arguments");
  hipLaunchKernelGGL(fold_copies_kernel, dim3((unsigned)n_entries), dim3(256), 0, (hipStream_t)stream, ws, table, copies, copy_stride);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                          void* dx, const void* add, float add_scale, float* dgamma, float* dbeta, float* partial,
                                          int accumulate, int64_t rows, int64_t D, int dt, void* dx_dropped, float drop_p,
                                          const uint64_t* rng_state, uint32_t rng_stream, int32_t acc_copies, int64_t acc_copy_stride,
                                          js2t_stream stream) {
  return js2t_layernorm_bwd_fused(dy, x, gamma, mean, rstd, dx, add, add_scale, dgamma, dbeta, partial, accumulate, rows, D, dt, dx_dropped,
                                  drop_p, rng_state, rng_stream, acc_copies, acc_copy_stride, nullptr, nullptr, stream);
}

extern "C" int js2t_layernorm_bwd_fused(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                        void* dx, const void* add, float add_scale, float* dgamma, float* dbeta, float* partial,
                                        int accumulate, int64_t rows, int64_t D, int dt, void* dx_dropped, float drop_p,
                                        const uint64_t* rng_state, uint32_t rng_stream, int32_t acc_copies, int64_t acc_copy_stride,
                                        const float* beta, void* n_out, js2t_stream stream) {
  const int copies = (accumulate && acc_copies > 1) ? acc_copies : 1;
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(dy && x && gamma && mean && rstd && dx && rows > 0 && D > 0, "layernorm_bwd: bad arguments");
  JS2T_CHECK(!dx_dropped || (rng_state && drop_p > 0.f && drop_p < 1.f), "layernorm_bwd_dropout: dx_dropped needs rng_state and 0 < p < 1");
  void* dxd = dx_dropped;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (D % 8 == 0) && D <= 64 * 8 * LNV_MAXCH &&
                   (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)add | (uintptr_t)dxd) & 15) == 0;
  JS2T_CHECK(vec || !dxd, "layernorm_bwd_dropout: the fused dropout output needs the vectorised kernel (D % 8 == 0, D <= 2048, 16-byte aligned)");
  JS2T_CHECK(!n_out || (vec && beta && (((uintptr_t)n_out | (uintptr_t)beta) & 15) == 0),
             "layernorm_bwd_fused: n_out needs beta and the vectorised kernel (D % 8 == 0, D <= 2048, 16-byte aligned)");
  if (vec) {
    const bool want_p = dgamma && dbeta;
    // += onto the gradient: atomics from the dx kernel itself - unless the deterministic switch is on (partial slab + the
    // fixed-order final kernel, which then adds onto the gradient)
    const bool direct = want_p && accumulate && !(g_js2t_deterministic && partial);
    JS2T_CHECK(!want_p || direct || partial, "layernorm_bwd: partial workspace required for dgamma/dbeta");
    // waves per block: as many as a 64 KB cross-wave reduction buffer allows (16 for D <= 512)
    int nw = 16;
    while (nw > 4 && (size_t)nw * 2 * D * sizeof(float) > 65536) nw >>= 1;
    // rows per wave: one block per CU is resident (16 waves x 100 VGPRs), so the rows are cut to fill ONE round of the
    // 256 CUs - 12000 rows: 3 per wave = 250 blocks (4 per wave left 68 CUs idle), 2592 rows: 1 per wave = 162 blocks
    // (were 41)
    int rpw = (int)((rows + 256 * (int64_t)nw - 1) / (256 * (int64_t)nw));
    rpw = rpw < 1 ? 1 : (rpw > LNV_ROWS / nw ? LNV_ROWS / nw : rpw);
    if (rpw * nw < LNV_MIN_ROWS) rpw = LNV_MIN_ROWS / nw;
    const int rows_per_block = rpw * nw;
    const int64_t nblk = (rows + rows_per_block - 1) / rows_per_block;
    const size_t lds = sizeof(float) * 2 * nw * D;
    float* part = (want_p && !direct) ? partial : (float*)nullptr;
    float* dga = direct ? dgamma : (float*)nullptr;
    float* dba = direct ? dbeta : (float*)nullptr;
    int rc;
    if (dt == JS2T_F32) {
      rc = D <= 512 ? launch_ln_bwd_vec<float, 1>(dy, x, gamma, mean, rstd, dx, add, add_scale, part, dga, dba, rows, (int)D, nblk, nw, lds, s, dxd, drop_p, rng_state, rng_stream, rows_per_block, copies, acc_copy_stride, beta, n_out)
         : D <= 1024 ? launch_ln_bwd_vec<float, 2>(dy, x, gamma, mean, rstd, dx, add, add_scale, part, dga, dba, rows, (int)D, nblk, nw, lds, s, dxd, drop_p, rng_state, rng_stream, rows_per_block, copies, acc_copy_stride, beta, n_out)
                     : launch_ln_bwd_vec<float, 4>(dy, x, gamma, mean, rstd, dx, add, add_scale, part, dga, dba, rows, (int)D, nblk, nw, lds, s, dxd, drop_p, rng_state, rng_stream, rows_per_block, copies, acc_copy_stride, beta, n_out);
    } else {
      rc = D <= 512 ? launch_ln_bwd_vec<uint16_t, 1>(dy, x, gamma, mean, rstd, dx, add, add_scale, part, dga, dba, rows, (int)D, nblk, nw, lds, s, dxd, drop_p, rng_state, rng_stream, rows_per_block, copies, acc_copy_stride, beta, n_out)
         : D <= 1024 ? launch_ln_bwd_vec<uint16_t, 2>(dy, x, gamma, mean, rstd, dx, add, add_scale, part, dga, dba, rows, (int)D, nblk, nw, lds, s, dxd, drop_p, rng_state, rng_stream, rows_per_block, copies, acc_copy_stride, beta, n_out)
                     : launch_ln_bwd_vec<uint16_t, 4>(dy, x, gamma, mean, rstd, dx, add, add_scale, part, dga, dba, rows, (int)D, nblk, nw, lds, s, dxd, drop_p, rng_state, rng_stream, rows_per_block, copies, acc_copy_stride, beta, n_out);
    }
    if (rc != JS2T_OK) return rc;
    JS2T_LAUNCH_CHECK();
    if (want_p && !direct) {
      hipLaunchKernelGGL(layernorm_bwd_param_final_kernel, dim3(cdiv(D, 64)), dim3(256), 0, s, partial, dgamma, dbeta, nblk, D,
                         accumulate);
      JS2T_LAUNCH_CHECK();
    }
    return JS2T_OK;
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((layernorm_bwd_dx_kernel<T>), dim3(cdiv(rows, ROWS_PER_BLOCK)), dim3(256), 0, s,
                                        (const T*)dy, (const T*)x, gamma, mean, rstd, (T*)dx, (const T*)add, add_scale,
                                        rows, D));
  JS2T_LAUNCH_CHECK();
  if (dgamma && dbeta) {
    JS2T_CHECK(partial, "layernorm_bwd: partial workspace required for dgamma/dbeta");
    const int64_t nparts = (rows + LN_SLAB - 1) / LN_SLAB;
    JS2T_CHECK(nparts <= 65535, "layernorm_bwd: too many rows");
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((layernorm_bwd_param_partial_kernel<T>), dim3(cdiv(D, 64), (unsigned)nparts),
                                          dim3(256), 0, s, (const T*)dy, (const T*)x, mean, rstd, partial, rows, D));
    JS2T_LAUNCH_CHECK();
    hipLaunchKernelGGL(layernorm_bwd_param_final_kernel, dim3(cdiv(D, 64)), dim3(256), 0, s, partial, dgamma, dbeta, nparts,
                       D, accumulate);
    JS2T_LAUNCH_CHECK();
  }
  return JS2T_OK;
}

extern "C" int js2t_softmax_fwd(const void* S, const uint8_t* mask, void* P, void* Pd, int64_t B, int64_t H, int64_t Tq,
                                int64_t Tk, int64_t ld, int64_t mask_sb, int64_t mask_sq, int dt, float p,
                                const uint64_t* rng_state, uint32_t rng_stream, js2t_stream stream) {
  if (B * H * Tq == 0) return JS2T_OK;
  JS2T_CHECK(S && P && Pd && Tk > 0 && ld >= Tk, "softmax_fwd: bad arguments");
  JS2T_CHECK(p >= 0.f && p < 1.f && (p == 0.f || (rng_state && Pd != P)), "softmax_fwd: bad dropout arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((softmax_fwd_kernel<T>), dim3(cdiv(B * H * Tq, ROWS_PER_BLOCK)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)S, mask, (T*)P, (T*)Pd, B, H, Tq, Tk, ld, mask_sb,
                                        mask_sq, p, rng_state, rng_stream));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_softmax_bwd(const void* P, const void* dPd, void* dS, int64_t Z, int64_t Tq, int64_t Tk, int64_t ld,
                                int dt, float p, const uint64_t* rng_state, uint32_t rng_stream, js2t_stream stream) {
  if (Z * Tq == 0) return JS2T_OK;
  JS2T_CHECK(P && dPd && dS && Tk > 0 && ld >= Tk, "softmax_bwd: bad arguments");
  JS2T_CHECK(p >= 0.f && p < 1.f && (p == 0.f || rng_state), "softmax_bwd: bad dropout arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3(cdiv(Z * Tq, ROWS_PER_BLOCK)), dim3(256), 0,
                                        (hipStream_t)stream, (const T*)P, (const T*)dPd, (T*)dS, Z * Tq, Tk, ld, p, rng_state,
                                        rng_stream));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_attn_head_mean(const void* P, float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t ld,
                                   int dt, js2t_stream stream) {
  if (B * Tq * Tk == 0) return JS2T_OK;
  JS2T_CHECK(P && out && H > 0, "attn_head_mean: bad arguments");
  int64_t g = (B * Tq * Tk + 255) / 256;
  if (g > 4096) g = 4096;
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((attn_head_mean_kernel<T>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                                        (const T*)P, out, B, H, Tq, Tk, ld));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_rel_bias_add(void* S, const float* rel_bias, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t ld,
                                 int32_t R, int dt, js2t_stream stream) {
  if (B * H * Tq * Tk == 0) return JS2T_OK;
  JS2T_CHECK(S && rel_bias && R >= 0 && ld >= Tk, "rel_bias_add: bad arguments");
  int64_t g = (B * H * Tq * Tk + 255) / 256;
  if (g > 4096) g = 4096;
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((rel_bias_add_kernel<T>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (T*)S,
                                        rel_bias, B, H, Tq, Tk, ld, (int)R));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

namespace {
__global__ void fixed_to_float_add_kernel(const long long* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n && src[i] != 0) dst[i] += (float)((double)src[i] * (1.0 / 4294967296.0) * (double)scale);
}
// deterministic form of rel_bias_grad_kernel: the same histogram in 2^-32 fixed point (integer atomics commute)
template <typename T>
__global__ void rel_bias_grad_fixed_kernel(const T* __restrict__ dS, unsigned long long* __restrict__ d_fix, int64_t B, int64_t H, int64_t Tq,
                                           int64_t Tk, int64_t ld, int R, int64_t rows_per_block) {
  extern __shared__ unsigned long long hist_i[];
  const int W = 2 * R + 1;
  const int64_t h = blockIdx.y;
  for (int i = threadIdx.x; i < W; i += blockDim.x) hist_i[i] = 0ull;
  __syncthreads();
  const int64_t r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, B * Tq);
  for (int64_t i = r0 * Tk + threadIdx.x; i < r1 * Tk; i += blockDim.x) {
    const int64_t bq = i / Tk, k = i - bq * Tk, b = bq / Tq, q = bq - b * Tq;
    const float g = io<T>::ld(dS + ((b * H + h) * Tq + q) * ld + k);
    const int64_t dlt = k - q;
    const int r = (int)(dlt < -R ? -R : (dlt > R ? R : dlt)) + R;
    if (g != 0.f) atomicAdd(&hist_i[r], js2t_to_fixed(g));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < W; i += blockDim.x)
    if (hist_i[i] != 0ull) atomicAdd(d_fix + h * W + i, hist_i[i]);
}
}  // namespace

// One buffer for the life of the process, allocated ONCE at its full size by the first call and never moved: hipGraphs captured
// later keep its address in their memset, kernel and conversion nodes (round 5 grew it by hipFree + hipMalloc when a request passed
// the capacity - graphs captured before then replayed onto freed memory, and growing inside a capture is an illegal hipMalloc;
// ADVICE r5).  A request beyond the capacity, or a FIRST call while `s` is capturing, fails (nullptr + the error string) instead.
// Single-stream: the deterministic launches that share it are ordered on the caller's stream.
constexpr size_t JS2T_FIXED_SCRATCH_WORDS = size_t(1) << 20;  // 8 MB: H (2 R + 1) of 128 heads at clip distance 4095
long long* js2t_fixed_scratch(size_t n, hipStream_t s) {
  static long long* buf = nullptr;
  if (n > JS2T_FIXED_SCRATCH_WORDS) {
    js2t_set_error("deterministic scratch: %zu words asked for, %zu there (heads x (2 R + 1) too large for the ordered histogram)", n,
                   JS2T_FIXED_SCRATCH_WORDS);
    return nullptr;
  }
  if (!buf) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      js2t_set_error("deterministic scratch: first use inside a hipGraph capture (run one eager step in this mode first)");
      return nullptr;
    }
    if (hipMalloc(&buf, JS2T_FIXED_SCRATCH_WORDS * sizeof(long long)) != hipSuccess) {
      buf = nullptr;
      js2t_set_error("deterministic scratch: hipMalloc of %zu bytes failed", JS2T_FIXED_SCRATCH_WORDS * sizeof(long long));
    }
  }
  return buf;
}
int js2t_fixed_to_float_add(const long long* src, float* dst, int64_t n, float scale, hipStream_t s) {
  hipLaunchKernelGGL(fixed_to_float_add_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, src, dst, n, scale);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_rel_bias_grad(const void* dS, float* d_rel_bias, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t ld,
                                  int32_t R, int dt, js2t_stream stream) {
  if (B * H * Tq * Tk == 0) return JS2T_OK;
  JS2T_CHECK(dS && d_rel_bias && R >= 0 && R <= 4096 && ld >= Tk && H <= 65535, "rel_bias_grad: bad arguments");
  const int64_t rows = B * Tq, rpb = 64;
  if (g_js2t_deterministic) {  // js2t_set_deterministic: fixed-point histogram, then one conversion pass
    const int64_t n = H * (2 * R + 1);
    long long* fix = js2t_fixed_scratch((size_t)n, (hipStream_t)stream);
    if (!fix) return JS2T_ERR_INVALID;  // (the error string says why)
    JS2T_CHECK(hipMemsetAsync(fix, 0, (size_t)n * sizeof(long long), (hipStream_t)stream) == hipSuccess, "rel_bias_grad: memset failed");
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((rel_bias_grad_fixed_kernel<T>), dim3((unsigned)cdiv(rows, rpb), (unsigned)H), dim3(256),
                                          (size_t)(2 * R + 1) * sizeof(unsigned long long), (hipStream_t)stream, (const T*)dS,
                                          (unsigned long long*)fix, B, H, Tq, Tk, ld, (int)R, rpb));
    JS2T_LAUNCH_CHECK();
    return js2t_fixed_to_float_add(fix, d_rel_bias, n, 1.f, (hipStream_t)stream);
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((rel_bias_grad_kernel<T>), dim3((unsigned)cdiv(rows, rpb), (unsigned)H), dim3(256),
                                        (size_t)(2 * R + 1) * sizeof(float), (hipStream_t)stream, (const T*)dS, d_rel_bias, B, H,
                                        Tq, Tk, ld, (int)R, rpb));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_fold_ln_weights(const int64_t* table, int32_t n_entries, int32_t max_rows, js2t_stream stream) {
  if (n_entries == 0 || max_rows == 0) return JS2T_OK;
  JS2T_CHECK(table && n_entries > 0 && n_entries <= 65535 && max_rows > 0, "fold_ln_weights: bad arguments");
  hipLaunchKernelGGL(fold_ln_weights_kernel, dim3((unsigned)cdiv(max_rows, 4), (unsigned)n_entries), dim3(256), 0, (hipStream_t)stream, table);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
