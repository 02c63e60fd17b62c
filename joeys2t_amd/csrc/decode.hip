// Single-query attention for incremental (KV-cached) decoding: one new target position per hypothesis attends to its own
// key/value history (self-attention) or to the encoder states of its utterance (cross-attention).
// The reference re-runs the whole decoder over the prefix at every step (search.py:518-534) and re-projects the encoder
// states for every layer and step; here keys / values are computed once and this kernel reads them in place:
//   * self-attention cache  kv[rows_phys, Tmax, ld]: hypothesis r's position j lives in physical row idx[r*idx_ld + j] -
//     beam re-ordering only rewrites that small table, the cache itself is never copied;
//   * cross-attention memory kv[n_utt, Tmax, ld]: hypothesis r reads utterance idx[r] (idx_ld == 0), with the utterance's
//     key-padding mask - the encoder states are not tiled beam-size times.
// One block per (hypothesis, head): scores by one wave per key (coalesced row reads), block softmax in LDS, output by one
// thread per head column.  f32 arithmetic, bf16 or f32 storage.  HBM/L2-bound: 2 * len * dh elements per block.
#include "common.hpp"

namespace {

constexpr int DEC_MAX_KEYS = 8192;

template <typename T>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ kbase,
                                                          const T* __restrict__ vbase, int64_t ldkv, const int32_t* __restrict__ idx,
                                                          int idx_ld, int Tmax, int len_host, const int32_t* __restrict__ len_dev,
                                                          const uint8_t* __restrict__ kmask,
                                                          T* __restrict__ out, int64_t ldo, int dh, float scale) {
  __shared__ float sc[DEC_MAX_KEYS];
  __shared__ float qs[256];
  __shared__ float red[8];
  const int r = blockIdx.x, h = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int len = len_dev ? min(max(*len_dev, 1), min(Tmax, DEC_MAX_KEYS)) : len_host;  // device-side count: replayed hipGraph
  if (t < dh) qs[t] = io<T>::ld(q + (int64_t)r * ldq + h * dh + t) * scale;  // q / sqrt(dh) BEFORE the product (reference :86-89)
  __syncthreads();
  const int64_t fixed_row = idx_ld == 0 ? (int64_t)idx[r] : 0;
  // scores: wave w takes keys w, w+4, ...
  for (int j = w; j < len; j += 4) {
    const int64_t prow = idx_ld == 0 ? fixed_row : (int64_t)idx[(int64_t)r * idx_ld + j];
    const T* kp = kbase + (prow * Tmax + j) * ldkv + h * dh;
    float a = 0.f;
    for (int c = lane; c < dh; c += 64) a += qs[c] * io<T>::ld(kp + c);
    a = wave_sum(a);
    if (lane == 0) sc[j] = (kmask && !kmask[prow * Tmax + j]) ? -INFINITY : a;
  }
  __syncthreads();
  // softmax over sc[0..len)
  float mx = -INFINITY;
  for (int j = t; j < len; j += 256) mx = fmaxf(mx, sc[j]);
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = t; j < len; j += 256) {
    const float e = __expf(sc[j] - mx);  // all keys masked: exp(nan) -> nan, as softmax over -inf
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.f / ((red[4] + red[5]) + (red[6] + red[7]));
  // context: thread c accumulates column c over the keys
  if (t < dh) {
    float acc = 0.f;
    for (int j = 0; j < len; ++j) {
      const int64_t prow = idx_ld == 0 ? fixed_row : (int64_t)idx[(int64_t)r * idx_ld + j];
      acc += sc[j] * io<T>::ld(vbase + (prow * Tmax + j) * ldkv + h * dh + t);
    }
    io<T>::st(out + (int64_t)r * ldo + h * dh + t, acc * inv);
  }
}


// Fast path: head size a multiple of the 16-byte vector (8 bf16 / 4 f32), aligned rows.  GROUP consecutive hypotheses
// that share their key / value rows (the beams of one utterance in cross-attention) are handled by ONE block, so the
// keys and values are read once per utterance and head instead of once per beam.
//  * scores: one thread per key, 16-byte loads along the head dimension (a lane walks its own cache line), the queries
//    are broadcast from LDS;
//  * context: threads = (key group, 16-byte column chunk); partial sums are combined by lane shuffles, then across the
//    four waves in LDS.
template <typename T> struct vec16;
template <> struct vec16<uint16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void ld(const uint16_t* p, float (&o)[8]) {
    const uint4 r = *(const uint4*)p;
    const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = __uint_as_float(rr[i] << 16);
      o[2 * i + 1] = __uint_as_float(rr[i] & 0xffff0000u);
    }
  }
};
template <> struct vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
    const float4 r = *(const float4*)p;
    o[0] = r.x, o[1] = r.y, o[2] = r.z, o[3] = r.w;
  }
};

constexpr int DEC_GROUP_MAX = 8;

template <typename T, int GROUP>
__global__ __launch_bounds__(256) void attn_decode_fast_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ kbase,
                                                               const T* __restrict__ vbase, int64_t ldkv, const int32_t* __restrict__ idx,
                                                               int idx_ld, int Tmax, int len_host, const int32_t* __restrict__ len_dev,
                                                               const uint8_t* __restrict__ kmask, T* __restrict__ out, int64_t ldo, int dh,
                                                               float scale, int rows) {
  constexpr int VN = vec16<T>::N;
  extern __shared__ float dsm[];  // qs[GROUP][dh] | sc[GROUP][len_pad] | red[4][GROUP][dh]
  const int len = len_dev ? min(max(*len_dev, 1), Tmax) : len_host;
  const int len_pad = (len + 3) & ~3;
  float* qs = dsm;
  float* sc = qs + GROUP * dh;
  float* red = sc + GROUP * len_pad;
  __shared__ float mred[GROUP][8];
  const int r0 = blockIdx.x * GROUP, h = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < GROUP * dh; i += 256) {
    const int g = i / dh, c = i - g * dh;
    qs[i] = r0 + g < rows ? io<T>::ld(q + (int64_t)(r0 + g) * ldq + h * dh + c) * scale : 0.f;
  }
  __syncthreads();
  const int64_t fixed_row = idx_ld == 0 ? (int64_t)idx[r0] : 0;  // GROUP > 1 only with idx_ld == 0 (shared rows)
  // ---- scores
  for (int j = t; j < len; j += 256) {
    const int64_t prow = idx_ld == 0 ? fixed_row : (int64_t)idx[(int64_t)r0 * idx_ld + j];
    const T* kp = kbase + (prow * Tmax + j) * ldkv + h * dh;
    float a[GROUP];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) a[g] = 0.f;
    for (int c = 0; c < dh; c += VN) {
      float kv[VN];
      vec16<T>::ld(kp + c, kv);
#pragma unroll
      for (int g = 0; g < GROUP; ++g)
#pragma unroll
        for (int e = 0; e < VN; ++e) a[g] += qs[g * dh + c + e] * kv[e];
    }
    const bool dead = kmask && !kmask[prow * Tmax + j];
#pragma unroll
    for (int g = 0; g < GROUP; ++g) sc[g * len_pad + j] = dead ? -INFINITY : a[g];
  }
  __syncthreads();
  // ---- softmax per query
  float inv[GROUP];
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    float mx = -INFINITY;
    for (int j = t; j < len; j += 256) mx = fmaxf(mx, sc[g * len_pad + j]);
    mx = wave_max(mx);
    if (lane == 0) mred[g][w] = mx;
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < GROUP; ++g) {
    const float mx = fmaxf(fmaxf(mred[g][0], mred[g][1]), fmaxf(mred[g][2], mred[g][3]));
    float sum = 0.f;
    for (int j = t; j < len; j += 256) {
      const float e = __expf(sc[g * len_pad + j] - mx);
      sc[g * len_pad + j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) mred[g][4 + w] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int g = 0; g < GROUP; ++g) inv[g] = 1.f / ((mred[g][4] + mred[g][5]) + (mred[g][6] + mred[g][7]));
  // ---- context: thread = (key group kg, column chunk tc)
  const int CT = dh / VN;            // chunks per row: power of two <= 64 (dispatcher)
  const int tc = t & (CT - 1), kg = t / CT, KG = 256 / CT;
  float o[GROUP][VN];
#pragma unroll
  for (int g = 0; g < GROUP; ++g)
#pragma unroll
    for (int e = 0; e < VN; ++e) o[g][e] = 0.f;
  for (int j = kg; j < len; j += KG) {
    const int64_t prow = idx_ld == 0 ? fixed_row : (int64_t)idx[(int64_t)r0 * idx_ld + j];
    float vv[VN];
    vec16<T>::ld(vbase + (prow * Tmax + j) * ldkv + h * dh + tc * VN, vv);
#pragma unroll
    for (int g = 0; g < GROUP; ++g) {
      const float pj = sc[g * len_pad + j];
#pragma unroll
      for (int e = 0; e < VN; ++e) o[g][e] += pj * vv[e];
    }
  }
  // lanes with equal tc inside a wave: offsets CT, 2CT, ... 32
#pragma unroll
  for (int g = 0; g < GROUP; ++g)
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      float v = o[g][e];
      for (int off = 32; off >= CT; off >>= 1) v += __shfl_xor(v, off, 64);
      o[g][e] = v;
    }
  if (lane < CT) {
#pragma unroll
    for (int g = 0; g < GROUP; ++g)
#pragma unroll
      for (int e = 0; e < VN; ++e) red[(w * GROUP + g) * dh + lane * VN + e] = o[g][e];
  }
  __syncthreads();
  for (int i = t; i < GROUP * dh; i += 256) {
    const int g = i / dh, c = i - g * dh;
    if (r0 + g < rows) {
      const float v = (red[(0 * GROUP + g) * dh + c] + red[(1 * GROUP + g) * dh + c]) + (red[(2 * GROUP + g) * dh + c] + red[(3 * GROUP + g) * dh + c]);
      io<T>::st(out + (int64_t)(r0 + g) * ldo + h * dh + c, v * inv[g]);
    }
  }
}

template <typename T, int GROUP>
int launch_decode_fast(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const int32_t* idx, int idx_ld, int Tmax,
                       int len, const int32_t* len_dev, const uint8_t* kmask, void* out, int64_t ldo, int rows, int H, int dh, float scale,
                       hipStream_t s) {
  const int len_cap = len_dev ? Tmax : len;
  const size_t lds = sizeof(float) * ((size_t)GROUP * dh + (size_t)GROUP * ((len_cap + 3) & ~3) + (size_t)4 * GROUP * dh);
  if (lds > 60 * 1024) return 1;  // caller falls back
  hipLaunchKernelGGL((attn_decode_fast_kernel<T, GROUP>), dim3((unsigned)cdiv(rows, GROUP), (unsigned)H), dim3(256), lds, s, (const T*)q,
                     ldq, (const T*)k, (const T*)v, ldkv, idx, idx_ld, Tmax, len, len_dev, kmask, (T*)out, ldo, dh, scale, rows);
  return 0;
}

}  // namespace

extern "C" int js2t_attn_decode(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const int32_t* idx, int32_t idx_ld,
                                int32_t Tmax, int32_t len, const int32_t* len_dev, const uint8_t* key_mask, void* out, int64_t ldo,
                                int32_t rows, int32_t H, int32_t dh, float scale, int32_t group, int dt, js2t_stream stream) {
  if (rows == 0 || H == 0) return JS2T_OK;
  JS2T_CHECK(q && k && v && idx && out, "attn_decode: null pointer");
  JS2T_CHECK(dh >= 1 && dh <= 256, "attn_decode: head size %d not in [1, 256]", dh);
  JS2T_CHECK(len_dev || (len >= 1 && len <= Tmax && len <= DEC_MAX_KEYS), "attn_decode: bad key count %d (Tmax %d, limit %d)", len, Tmax,
             DEC_MAX_KEYS);
  JS2T_CHECK(H <= 65535, "attn_decode: too many heads");
  JS2T_CHECK(group >= 1 && (group == 1 || idx_ld == 0), "attn_decode: grouped hypotheses must share their key rows (idx_ld == 0)");
  {
    // fast path: 16-byte vectors along the head dimension, `group` hypotheses per block
    const int vn = dt == JS2T_F32 ? 4 : 8, es = dt == JS2T_F32 ? 4 : 2;
    const int ct = dh / vn;
    const bool pow2 = ct >= 1 && ct <= 64 && (ct & (ct - 1)) == 0;
    const bool aligned = dh % vn == 0 && ((ldkv * es) % 16) == 0 && ((((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0 && ((dh * es) % 16) == 0;
    if (pow2 && aligned) {
      const int g = (group == 5 || group == 4 || group == 2 || group == 8) ? group : 1;
      hipStream_t s = (hipStream_t)stream;
      int rc = 1;
#define DEC_FAST(T, G) rc = launch_decode_fast<T, G>(q, ldq, k, v, ldkv, idx, idx_ld, Tmax, len, len_dev, key_mask, out, ldo, rows, H, dh, scale, s)
      if (dt == JS2T_F32) {
        if (g == 8) DEC_FAST(float, 8); else if (g == 5) DEC_FAST(float, 5); else if (g == 4) DEC_FAST(float, 4); else if (g == 2) DEC_FAST(float, 2); else DEC_FAST(float, 1);
      } else {
        if (g == 8) DEC_FAST(uint16_t, 8); else if (g == 5) DEC_FAST(uint16_t, 5); else if (g == 4) DEC_FAST(uint16_t, 4); else if (g == 2) DEC_FAST(uint16_t, 2); else DEC_FAST(uint16_t, 1);
      }
#undef DEC_FAST
      if (rc == 0) {
        JS2T_LAUNCH_CHECK();
        return JS2T_OK;
      }
    }
  }
  if (dt == JS2T_F32) {
    hipLaunchKernelGGL((attn_decode_kernel<float>), dim3((unsigned)rows, (unsigned)H), dim3(256), 0, (hipStream_t)stream, (const float*)q, ldq,
                       (const float*)k, (const float*)v, ldkv, idx, idx_ld, Tmax, len, len_dev, key_mask, (float*)out, ldo, dh, scale);
  } else {
    hipLaunchKernelGGL((attn_decode_kernel<uint16_t>), dim3((unsigned)rows, (unsigned)H), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)q, ldq, (const uint16_t*)k, (const uint16_t*)v, ldkv, idx, idx_ld, Tmax, len, len_dev, key_mask,
                       (uint16_t*)out, ldo, dh, scale);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
