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
                                                          int idx_ld, int Tmax, int len, const uint8_t* __restrict__ kmask,
                                                          T* __restrict__ out, int64_t ldo, int dh, float scale) {
  __shared__ float sc[DEC_MAX_KEYS];
  __shared__ float qs[256];
  __shared__ float red[8];
  const int r = blockIdx.x, h = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
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

}  // namespace

extern "C" int js2t_attn_decode(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const int32_t* idx, int32_t idx_ld,
                                int32_t Tmax, int32_t len, const uint8_t* key_mask, void* out, int64_t ldo, int32_t rows, int32_t H,
                                int32_t dh, float scale, int dt, js2t_stream stream) {
  if (rows == 0 || H == 0) return JS2T_OK;
  JS2T_CHECK(q && k && v && idx && out, "attn_decode: null pointer");
  JS2T_CHECK(dh >= 1 && dh <= 256, "attn_decode: head size %d not in [1, 256]", dh);
  JS2T_CHECK(len >= 1 && len <= Tmax && len <= DEC_MAX_KEYS, "attn_decode: bad key count %d (Tmax %d, limit %d)", len, Tmax, DEC_MAX_KEYS);
  JS2T_CHECK(H <= 65535, "attn_decode: too many heads");
  if (dt == JS2T_F32) {
    hipLaunchKernelGGL((attn_decode_kernel<float>), dim3((unsigned)rows, (unsigned)H), dim3(256), 0, (hipStream_t)stream, (const float*)q, ldq,
                       (const float*)k, (const float*)v, ldkv, idx, idx_ld, Tmax, len, key_mask, (float*)out, ldo, dh, scale);
  } else {
    hipLaunchKernelGGL((attn_decode_kernel<uint16_t>), dim3((unsigned)rows, (unsigned)H), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)q, ldq, (const uint16_t*)k, (const uint16_t*)v, ldkv, idx, idx_ld, Tmax, len, key_mask,
                       (uint16_t*)out, ldo, dh, scale);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
