// Shared device helpers for the joeys2t_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/joeys2t_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---------------------------------------------------------------- error plumbing
void js2t_set_error(const char* fmt, ...);
#define JS2T_CHECK(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      js2t_set_error(__VA_ARGS__);       \
      return JS2T_ERR_INVALID;           \
    }                                    \
  } while (0)
#define JS2T_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      js2t_set_error("launch failed: %s", hipGetErrorString(e__));       \
      return JS2T_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

// Per-caller settings (core.cpp, js2t_ctx_* of the header): the value of `key` for the launch being made - the process-wide test
// override if one is set, else the context bound to the calling thread (js2t_ctx_bind), else the built-in default.
int js2t_ctx_value(int key);
void js2t_ctx_override(int key, int value);  // what the process-wide setters (js2t_set_deterministic, js2t_gemm_*_mode) write
// ordered sums instead of floating-point atomics (js2t_ctx_set(ctx, JS2T_CTX_DETERMINISTIC, 1), or the test override js2t_set_deterministic)
#define g_js2t_deterministic js2t_ctx_value(JS2T_CTX_DETERMINISTIC)

// Deterministic mode of the histogram-shaped sums (relative-position bias gradient): the addends are rounded ONCE to 2^-32 fixed
// point and summed as 64-bit integers - integer atomics commute, so LDS and global atomics in any order give the same bits.
// js2t_fixed_scratch: a device buffer of n zeroed-by-the-caller int64 words owned by the library (allocated at first use - which
// must not be inside a hipGraph capture: the first step of a run is eager); js2t_fixed_to_float_add: dst[i] += src[i] * 2^-32 * scale.
constexpr float JS2T_FIX_SCALE = 4294967296.0f;
long long* js2t_fixed_scratch(size_t n, hipStream_t s);
int js2t_fixed_to_float_add(const long long* src, float* dst, int64_t n, float scale, hipStream_t s);
__device__ __forceinline__ unsigned long long js2t_to_fixed(float v) { return (unsigned long long)__float2ll_rn(v * JS2T_FIX_SCALE); }

// ---------------------------------------------------------------- dtype helpers
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950: RNE, NaN stays NaN
  return __builtin_bit_cast(uint16_t, h);
}

// two values -> one dword of bf16 bits (lo in the low half): ONE v_cvt_pk_bf16_f32.  Written as two scalar conversions joined by
// shift / or, the compiler converts with a wasted half each and spends two or three more VALU instructions putting the halves together.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t_;
  const f32x2_t_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t_));
}

template <typename T> struct io;
template <> struct io<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<uint16_t> {  // bf16 carried as raw bits
  static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }
};

// generic element load/store by runtime dtype code
__device__ __forceinline__ float ld_elem(const void* p, int64_t i, int dt) {
  return dt == JS2T_F32 ? ((const float*)p)[i] : bf16_bits_to_f32(((const uint16_t*)p)[i]);
}
__device__ __forceinline__ void st_elem(void* p, int64_t i, int dt, float v) {
  if (dt == JS2T_F32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = f32_to_bf16_bits(v);
}

// ---------------------------------------------------------------- LayerNorm-fold weights: rounding with error feedback
// Four centred, gamma-scaled weights w*g - cs -> bf16, the rounding error of each carried into the next (and, through `carry`,
// into the lane's next four).  The consuming product cancels the row mean of its activations through these rows summing to 0;
// rounded independently their sum is the sum of K rounding errors (~2e-3 for K = 512), which leaks rstd * mean * that into every
// output - 4 % of the output's scale for a row 30 sigma off zero (tests/test_hip_ln_fold.py).  With the feedback a lane's eight
// values sum to their exact sum within one last carry.  Shared by js2t_fold_ln_weights and js2t_adamw_items (bit-identical).
__device__ __forceinline__ uint2 ln_fold_round4(const float (&w)[4], const float4 g, float cs, float& carry) {
#pragma clang fp contract(off)
  const float gg[4] = {g.x, g.y, g.z, g.w};
  uint16_t q[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float target = __builtin_fmaf(w[i], gg[i], -cs) + carry;
    q[i] = f32_to_bf16_bits(target);
    carry = target - bf16_bits_to_f32(q[i]);
  }
  return make_uint2((uint32_t)q[0] | ((uint32_t)q[1] << 16), (uint32_t)q[2] | ((uint32_t)q[3] << 16));
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
// All-lanes butterfly reductions in the VALU: __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0), i.e. six LDS
// round trips per reduction on the critical path of every row-wise kernel (LayerNorm: two per row).  Inside a row of 16
// lanes DPP does the exchange (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: each a symmetric pairing, so every
// lane ends with the row's value); across the four rows gfx950's v_permlane16_swap / v_permlane32_swap trade whole rows.
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));   // quad_perm [1,0,3,2]
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));   // quad_perm [2,3,0,1]
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));  // row_half_mirror
  v = op(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));  // row_mirror
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = op(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return op(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  return wave_reduce(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
  return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); });
}
// block-wide sum; `red` is a __shared__ float[>= blockDim/64]; all threads get the result
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// LDS-DMA request written as inline assembly.  With the builtin the compiler knows that LDS is being written behind its
// back and puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 it cannot prove disjoint (the transposing read
// is an intrinsic without usable alias information; plain loads are not affected) - which drains the prefetch that the
// ring exists for.  The kernels order DMA and reads themselves (counted vmcnt + barrier), so the request is hidden:
// compiler-generated vmcnt waits for its own loads can only become stricter by the extra in-order entries, never weaker.
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(l) : "m0");
}

// bijective XCD-aware remap of a linear block id (blocks b and b+8 share an XCD): every XCD gets a contiguous range of
// the logical ids, so work items that read the same operand panel (neighbouring GEMM tiles, the query tiles of one
// attention head) hit in one L2 instead of pulling the panel through the fabric once per XCD
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---------------------------------------------------------------- counter-based dropout RNG
// Stateless: the keep decision of element (row, col) is a pure function of {seed, offset, call-site stream, row,
// col}, so forward and backward kernels regenerate identical masks and nothing is stored.  One 32-bit integer hash
// (two v_mul_lo_u32; "lowbias32") yields two 16-bit uniform values = two decisions, i.e. ~1 multiply per element —
// the first version used Philox4x32-7 (3.5 quarter-rate multiplies per element) and cost ~45% of an FFN epilogue.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// Second level of the dropout RNG: the word of (row, column pair) from the row's key, rowkey = hash32(row ^ key).  The row key is
// fully mixed already and the words of a row are rowkey + 0, 1, 2, ...: multiply - fold - multiply without the two outer folds of
// hash32 decorrelates them (two vector instructions fewer per word).  Chosen by measurement, not by taste: ONE multiply round
// (fold, multiply, fold) leaves neighbouring words anti-correlated at hundreds of sigma on a 4096 x 4096 mask; this form and
// hash32 itself stay within 3 sigma on rates and on every neighbour correlation (tests/test_hip_ops.py::test_dropout_rng_statistics,
// and its numpy twin tools/rng_stats.py for other keys and rates).
__device__ __forceinline__ uint32_t hash32w(uint32_t x) {
  x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  return x;
}
// The same word from xm = x * HASH32W_M1 (mod 2^32), for callers whose x advances by constants: the first multiply distributes
// over the sum (rowkey + counter) * M1 = rowkey * M1 + counter * M1, so a loop keeps rowkey * M1 and adds multiples of M1 - one
// multiply per word instead of two (attention.hip).  last_mul: HASH32W_M2 gives hash32w(x); HASH32W_M2 << 16 gives
// hash32w(x) << 16, i.e. the LOW half of the word in the upper 16 bits - for a lane that only ever wants one half of its words.
constexpr uint32_t HASH32W_M1 = 0x7feb352du, HASH32W_M2 = 0x846ca68bu;
__device__ __forceinline__ uint32_t hash32w_pre(uint32_t xm, uint32_t last_mul = HASH32W_M2) {
  xm ^= xm >> 15;
  return xm * last_mul;
}
__device__ __forceinline__ uint32_t dropout_key(const uint64_t* rng_state, uint32_t stream) {
  const uint64_t seed = rng_state[0], off = rng_state[1];
  return hash32((uint32_t)seed ^ hash32((uint32_t)(seed >> 32) + 0x9E3779B9u) ^ hash32((uint32_t)off * 0x85EBCA6Bu + 1u) ^
                hash32(((uint32_t)(off >> 32) ^ stream) * 0xC2B2AE35u + 2u));
}
// Keep-decisions for 4 consecutive columns (col4*4 .. col4*4+3) of `row`; bit i = keep column col4*4+i.
// P(drop) = floor(p * 65536) / 65536.
// `key` = dropout_key(...) is call-invariant: kernels compute it ONCE per thread and pass it in.
__device__ __forceinline__ uint32_t dropout_keep4_key(uint32_t key, uint32_t row, uint32_t col4, float p) {
  const uint32_t rowkey = hash32(row ^ key);
  const uint32_t h0 = hash32w(rowkey + 2u * col4), h1 = hash32w(rowkey + 2u * col4 + 1u);
  const uint32_t thr = (uint32_t)(p * 65536.0f);
  return ((h0 & 0xffffu) >= thr ? 1u : 0u) | ((h0 >> 16) >= thr ? 2u : 0u) | ((h1 & 0xffffu) >= thr ? 4u : 0u) |
         ((h1 >> 16) >= thr ? 8u : 0u);
}
__device__ __forceinline__ uint32_t dropout_keep4(const uint64_t* rng_state, uint32_t stream, uint32_t row, uint32_t col4,
                                                  float p) {
  return dropout_keep4_key(dropout_key(rng_state, stream), row, col4, p);
}
__device__ __forceinline__ bool dropout_keep1(const uint64_t* rng_state, uint32_t stream,
                                              uint32_t row, uint32_t col, float p) {
  return (dropout_keep4(rng_state, stream, row, col >> 2, p) >> (col & 3)) & 1u;
}

// ---------------------------------------------------------------- activations
__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case JS2T_ACT_RELU: return fmaxf(v, 0.f);
    case JS2T_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case JS2T_ACT_SWISH: return v / (1.f + __expf(-v));
    case JS2T_ACT_TANH: return tanhf(v);
    case JS2T_ACT_HARDSWISH: return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    default: return v;
  }
}
__device__ __forceinline__ float act_grad(float z, int act) {  // d act(z) / dz
  switch (act) {
    case JS2T_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case JS2T_ACT_GELU: {
      const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * z * z);
      return cdf + z * pdf;
    }
    case JS2T_ACT_SWISH: {
      const float s = 1.f / (1.f + __expf(-z));
      return s * (1.f + z * (1.f - s));
    }
    case JS2T_ACT_TANH: {
      const float t = tanhf(z);
      return 1.f - t * t;
    }
    case JS2T_ACT_HARDSWISH: return z < -3.f ? 0.f : (z > 3.f ? 1.f : (2.f * z + 3.f) * (1.f / 6.f));
    default: return 1.f;
  }
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
