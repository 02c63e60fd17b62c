// Loss-side kernels: row log-sum-exp / log-softmax over the vocabulary, label-smoothed cross-entropy
// (closed form — the [N,V] smoothed target of loss.py:35-58 is never materialised), CTC alpha/beta
// recursions in log space and the CTC gradient w.r.t. the logits, plus small deterministic reductions.
// All HBM-bound: one pass over the logits per kernel, wavefront-shuffle + LDS block reductions.
#include "common.hpp"

namespace {

constexpr int LB = 256;  // threads per row-block

__device__ __forceinline__ float logaddexp(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b);
  return m + log1pf(__expf(-fabsf(a - b)));
}

// the same with hardware exp / log (v_exp_f32, v_log_f32): log(1 + x) loses x below 6e-8, an absolute error far under the
// 1e-4 relative budget of the accumulated log-likelihood; used where the function sits on a serial dependency chain
__device__ __forceinline__ float logaddexp_fast(float a, float b) {
  const float m = fmaxf(a, b);
  const float d = -fabsf(a - b);  // nan when both are -inf
  const float r = m + __logf(1.f + __expf(d));
  return (m == -INFINITY) ? -INFINITY : r;
}

// ---------------------------------------------------------------- row statistics over V
// lse[r] = log sum_v exp(x[r,v]); argmax[r] = first index of the row maximum (optional).
template <typename T>
__global__ __launch_bounds__(LB) void row_lse_kernel(const T* __restrict__ x, float* __restrict__ lse,
                                                     int64_t* __restrict__ argmax, int64_t rows, int64_t V) {
  __shared__ float red[LB / 64];
  __shared__ int redi[LB / 64];
  const int64_t r = blockIdx.x;
  const T* xr = x + r * V;
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  for (int64_t v = threadIdx.x; v < V; v += LB) {
    const float f = io<T>::ld(xr + v);
    if (f > mx) { mx = f; mi = (int)v; }
  }
  const float bmx = block_max(mx, red);
  float s = 0.f;
  for (int64_t v = threadIdx.x; v < V; v += LB) s += __expf(io<T>::ld(xr + v) - bmx);
  s = block_sum(s, red);
  if (argmax) {
    int cand = (mx == bmx) ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cand;
    __syncthreads();
    if (threadIdx.x == 0) {
      int best = redi[0];
      for (int i = 1; i < LB / 64; ++i) best = min(best, redi[i]);
      argmax[r] = best;
    }
  }
  if (threadIdx.x == 0) lse[r] = bmx + __logf(s);
}

// bf16 rows of V % 8 == 0 (16-byte aligned): one 16-byte load per 8 logits, the row stays in registers between the
// maximum and the sum pass (one read of the logits instead of two scalar ones)
constexpr int RL_CH = 4;  // chunks of 8 per thread: V <= 8 * LB * RL_CH = 8192
__global__ __launch_bounds__(LB) void row_lse_vec_kernel(const uint16_t* __restrict__ x, float* __restrict__ lse,
                                                         int64_t* __restrict__ argmax, int64_t rows, int64_t V) {
  __shared__ float red[LB / 64];
  __shared__ int redi[LB / 64];
  const int64_t r = blockIdx.x;
  const uint4* xr = (const uint4*)(x + r * V);
  const int nch = (int)(V >> 3);
  float v[RL_CH][8];
  float mx = -INFINITY;
  int mi = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < RL_CH; ++j) {
    const int c = threadIdx.x + LB * j;
    if (c < nch) {
      const uint4 q = xr[c];
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[j][2 * i] = __uint_as_float(w[i] << 16);
        v[j][2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (v[j][i] > mx) { mx = v[j][i]; mi = 8 * c + i; }  // ascending index inside a thread: first maximum kept
    }
  }
  const float bmx = block_max(mx, red);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < RL_CH; ++j) {
    const int c = threadIdx.x + LB * j;
    if (c < nch) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s += __expf(v[j][i] - bmx);
    }
  }
  s = block_sum(s, red);
  if (argmax) {
    int cand = (mx == bmx) ? mi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cand;
    __syncthreads();
    if (threadIdx.x == 0) {
      int best = redi[0];
      for (int i = 1; i < LB / 64; ++i) best = min(best, redi[i]);
      argmax[r] = best;
    }
  }
  if (threadIdx.x == 0) lse[r] = bmx + __logf(s);
}

template <typename T, typename TO>
__global__ __launch_bounds__(LB) void log_softmax_kernel(const T* __restrict__ x, TO* __restrict__ y, int64_t rows, int64_t V) {
  __shared__ float red[LB / 64];
  const int64_t r = blockIdx.x;
  const T* xr = x + r * V;
  float mx = -INFINITY;
  for (int64_t v = threadIdx.x; v < V; v += LB) mx = fmaxf(mx, io<T>::ld(xr + v));
  mx = block_max(mx, red);
  float s = 0.f;
  for (int64_t v = threadIdx.x; v < V; v += LB) s += __expf(io<T>::ld(xr + v) - mx);
  s = block_sum(s, red);
  const float lse = mx + __logf(s);
  for (int64_t v = threadIdx.x; v < V; v += LB) io<TO>::st(y + r * V + v, io<T>::ld(xr + v) - lse);
}

// ---------------------------------------------------------------- label-smoothed cross-entropy
// Per row r with gold g = trg[r] (loss.py:35-58,100):  target t[v] = 1-eps at g, eps/(V-2) elsewhere, 0 at pad,
// whole row 0 when g == pad;  loss_r = sum_v t (log t - lp[v]).  eps <= 0: NLL with ignore_index = pad.
template <typename T>
__global__ __launch_bounds__(LB) void xent_fwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ trg,
                                                      float* __restrict__ loss_rows, float* __restrict__ correct_rows,
                                                      float* __restrict__ lse_out, int64_t rows, int64_t V, int64_t pad,
                                                      float eps) {
  __shared__ float red[LB / 64];
  __shared__ int redi[LB / 64];
  const int64_t r = blockIdx.x;
  const T* xr = x + r * V;
  const int64_t g = trg[r];
  float mx = -INFINITY;
  int mi = 0x7fffffff;
  float sx = 0.f;
  for (int64_t v = threadIdx.x; v < V; v += LB) {
    const float f = io<T>::ld(xr + v);
    sx += f;
    if (f > mx) { mx = f; mi = (int)v; }
  }
  const float bmx = block_max(mx, red);
  sx = block_sum(sx, red);
  float s = 0.f;
  for (int64_t v = threadIdx.x; v < V; v += LB) s += __expf(io<T>::ld(xr + v) - bmx);
  s = block_sum(s, red);
  int cand = (mx == bmx) ? mi : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cand;
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = redi[0];
    for (int i = 1; i < LB / 64; ++i) best = min(best, redi[i]);
    const float lse = bmx + __logf(s);
    lse_out[r] = lse;
    float loss = 0.f, ok = 0.f;
    if (g != pad && g >= 0 && g < V) {
      const float lpg = io<T>::ld(xr + g) - lse;
      if (eps > 0.f) {
        const float u = eps / (float)(V - 2);
        const float lpp = (pad >= 0 && pad < V) ? io<T>::ld(xr + pad) - lse : 0.f;
        const float sum_lp = sx - (float)V * lse;
        const float cent = (1.f - eps) * __logf(1.f - eps) + eps * __logf(u);
        loss = cent - ((1.f - eps) * lpg + u * (sum_lp - lpg - lpp));
      } else {
        loss = -lpg;
      }
      ok = (best == (int)g) ? 1.f : 0.f;
    }
    loss_rows[r] = loss;
    correct_rows[r] = ok;
  }
}

// The same on f32 logits with 16-byte rows (V % 4 == 0), ONE WAVE per row and four rows per block (round 5): a single pass of 16-byte
// loads with a running maximum per lane (online softmax: the partial sum is rescaled when the lane's maximum moves), wave reductions
// only - the block form above reads the row twice with 4-byte loads and crosses five block barriers per row (36 us for 2592 x 5000
// logits = 0.18 of the HBM rate).  The sums run in another order than above: the same loss to the last bits of an f32.
__global__ __launch_bounds__(256) void xent_fwd_wave_kernel(const float* __restrict__ x, const int64_t* __restrict__ trg,
                                                            float* __restrict__ loss_rows, float* __restrict__ correct_rows,
                                                            float* __restrict__ lse_out, int64_t rows, int64_t V, int64_t pad, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;  // (no barriers below)
  const float* xr = x + r * V;
  const int n4 = (int)(V >> 2);
  float m = -INFINITY, s = 0.f, sx = 0.f;
  int mi = 0x7fffffff;
  auto take = [&](const float4& q, int c) {
    const float m4 = fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w));
    if (m4 > m) {  // a strictly larger value: the lane's first index of its maximum
      mi = 4 * c + (q.x == m4 ? 0 : q.y == m4 ? 1 : q.z == m4 ? 2 : 3);
      s *= __expf(m - m4);  // (m = -inf: s is 0 and stays 0)
      m = m4;
    }
    s += (__expf(q.x - m) + __expf(q.y - m)) + (__expf(q.z - m) + __expf(q.w - m));
    sx += (q.x + q.y) + (q.z + q.w);
  };
  int c = lane;
  for (; c + 448 < n4; c += 512) {  // eight 16-byte loads in flight per lane
    float4 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = ((const float4*)xr)[c + 64 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) take(q[i], c + 64 * i);
  }
  for (; c + 192 < n4; c += 256) {
    const float4 q0 = ((const float4*)xr)[c], q1 = ((const float4*)xr)[c + 64], q2 = ((const float4*)xr)[c + 128], q3 = ((const float4*)xr)[c + 192];
    take(q0, c), take(q1, c + 64), take(q2, c + 128), take(q3, c + 192);
  }
  for (; c < n4; c += 64) take(((const float4*)xr)[c], c);
  const float bmx = wave_max(m);
  s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - bmx));
  sx = wave_sum(sx);
  int cand = (m == bmx) ? mi : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  if (lane == 0) {
    const int64_t g = trg[r];
    const float lse = bmx + __logf(s);
    lse_out[r] = lse;
    float loss = 0.f, ok = 0.f;
    if (g != pad && g >= 0 && g < V) {
      const float lpg = xr[g] - lse;
      if (eps > 0.f) {
        const float u = eps / (float)(V - 2);
        const float lpp = (pad >= 0 && pad < V) ? xr[pad] - lse : 0.f;
        const float sum_lp = sx - (float)V * lse;
        const float cent = (1.f - eps) * __logf(1.f - eps) + eps * __logf(u);
        loss = cent - ((1.f - eps) * lpg + u * (sum_lp - lpg - lpp));
      } else {
        loss = -lpg;
      }
      ok = (cand == (int)g) ? 1.f : 0.f;
    }
    loss_rows[r] = loss;
    correct_rows[r] = ok;
  }
}
// backward on the same rows: nothing to reduce, so ONE THREAD per eight logits over the whole [rows, V] array (two 16-byte loads, one or
// two 16-byte stores, the row's gold index and log-sum-exp from cache) - 25 waves per SIMD's worth of independent requests instead of one
// wave walking a row
template <typename TO>
__global__ __launch_bounds__(256) void xent_bwd_flat_kernel(const float* __restrict__ x, const int64_t* __restrict__ trg,
                                                            const float* __restrict__ lse, const float* __restrict__ g_dev, float scale,
                                                            TO* __restrict__ dx, int rows, int n8, int64_t V, int64_t pad, float eps) {
  const int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (chunk >= (int64_t)rows * n8) return;
  const int r = (int)(chunk / n8), c = (int)(chunk - (int64_t)r * n8);
  const int64_t g = trg[r];
  const bool dead = g == pad || g < 0 || g >= V;
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (!dead) {
    const float4 qa = ((const float4*)x)[2 * chunk], qb = ((const float4*)x)[2 * chunk + 1];
    const float gs = scale * (g_dev ? *g_dev : 1.f), l = lse[r];
    const float u = eps > 0.f ? eps / (float)(V - 2) : 0.f, tg = eps > 0.f ? 1.f - eps : 1.f;
    const float f[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    const int gi = (int)g - 8 * c, pi = (pad >= 0 && pad < V) ? (int)pad - 8 * c : -1;  // positions of gold / pad inside this chunk (or outside 0..7)
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = gs * (__expf(f[i] - l) - (i == gi ? tg : (i == pi ? 0.f : u)));
  }
  if constexpr (sizeof(TO) == 4) {
    ((float4*)dx)[2 * chunk] = make_float4(o[0], o[1], o[2], o[3]);
    ((float4*)dx)[2 * chunk + 1] = make_float4(o[4], o[5], o[6], o[7]);
  } else {
    ((uint4*)dx)[chunk] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
}

// dx[r,v] = gscale * (softmax - t)   (0 for pad rows);  gscale = scale * (*g_dev)
// TO: the gradient's storage type - the logits' own, or bf16 for f32 logits whose gradient feeds a bf16 product next
// (js2t_xent_bwd_as: no f32 gradient to write and cast afterwards)
template <typename T, typename TO = T>
__global__ __launch_bounds__(LB) void xent_bwd_kernel(const T* __restrict__ x, const int64_t* __restrict__ trg,
                                                      const float* __restrict__ lse, const float* __restrict__ g_dev,
                                                      float scale, TO* __restrict__ dx, int64_t rows, int64_t V, int64_t pad,
                                                      float eps) {
  const int64_t r = blockIdx.x;
  const int64_t g = trg[r];
  const float gs = scale * (g_dev ? *g_dev : 1.f);
  const T* xr = x + r * V;
  TO* dr = dx + r * V;
  if (g == pad || g < 0 || g >= V) {
    for (int64_t v = threadIdx.x; v < V; v += LB) io<TO>::st(dr + v, 0.f);
    return;
  }
  const float l = lse[r];
  const float u = eps > 0.f ? eps / (float)(V - 2) : 0.f;
  const float tg = eps > 0.f ? 1.f - eps : 1.f;
  for (int64_t v = threadIdx.x; v < V; v += LB) {
    const float sm = __expf(io<T>::ld(xr + v) - l);
    const float t = (v == g) ? tg : ((v == pad) ? 0.f : u);
    io<TO>::st(dr + v, gs * (sm - t));
  }
}

// ---------------------------------------------------------------- deterministic small reductions
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

// ---------------------------------------------------------------- CTC
// Extended label sequence of utterance b: ext(s) = blank for even s, targets[b, s/2] for odd s; S = 2L+1.
// Emission lp_t(s) = x[b,t,ext(s)] - lse[b,t].  alpha/beta are stored as f32 [B, T, Smax].
constexpr int CTC_THREADS = 256;
constexpr int CTC_MAX_S = 1024;
constexpr int CTC_CHUNK_FLOATS = 8192;  // emission staging: 32 KB of LDS

template <typename T, bool BACKWARD>
__device__ __forceinline__ void ctc_recursion_body(
    float (&a)[2][CTC_MAX_S + 2], int (&ext)[CTC_MAX_S], float (&em)[CTC_CHUNK_FLOATS],
    const T* __restrict__ x, const float* __restrict__ lse, const int64_t* __restrict__ targets,
    const int64_t* __restrict__ in_len, const int64_t* __restrict__ tgt_len, float* __restrict__ out /*alpha|beta*/,
    float* __restrict__ nll, int64_t Tmax, int64_t V, int64_t Lmax, int64_t Smax, int64_t blank, const int32_t* __restrict__ rowoff) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t Tb = min(in_len[b], Tmax);
  const int L = (int)min(tgt_len[b], Lmax);
  const int S = 2 * L + 1;
  const int64_t r0 = rowoff ? (int64_t)rowoff[b] : (int64_t)b * Tmax;  // first logits / lse row of the utterance (packed rows: js2t_ctc_alpha)
  for (int s = tid; s < S; s += CTC_THREADS) ext[s] = (s & 1) ? (int)targets[b * Lmax + (s >> 1)] : (int)blank;
  float* ob = out + (int64_t)b * Tmax * Smax;
  // rows t >= Tb are never read by the gradient kernel; keep them defined
  for (int64_t i = Tb * Smax + tid; i < Tmax * Smax; i += CTC_THREADS) ob[i] = -INFINITY;
  __syncthreads();
  if (Tb <= 0) {
    if (!BACKWARD && tid == 0) nll[b] = INFINITY;
    return;
  }
  const int chunk_T = max(1, min(64, CTC_CHUNK_FLOATS / S));
  int cur = 0;
  for (int64_t c0 = 0; c0 < Tb; c0 += chunk_T) {
    const int nt = (int)min((int64_t)chunk_T, Tb - c0);
    // stage emissions of this chunk of time steps (gather from the logits row)
    for (int i = tid; i < nt * S; i += CTC_THREADS) {
      const int tt = i / S, s = i - tt * S;
      const int64_t t = BACKWARD ? (Tb - 1 - (c0 + tt)) : (c0 + tt);
      const int lab = ext[s];
      const float v = (lab >= 0 && lab < V) ? io<T>::ld(x + (r0 + t) * V + lab) - lse[r0 + t] : -INFINITY;
      em[i] = v;
    }
    __syncthreads();
    for (int tt = 0; tt < nt; ++tt) {
      const int64_t step = c0 + tt;  // 0 .. Tb-1 in recursion order
      const int64_t t = BACKWARD ? (Tb - 1 - step) : step;
      const float* prev = a[cur ^ 1];
      float* now = a[cur];
      for (int s = tid; s < S; s += CTC_THREADS) {
        const float e = em[tt * S + s];
        float v;
        if (step == 0) {
          const bool start = BACKWARD ? (s >= S - 2) : (s < 2);
          v = start ? e : -INFINITY;
        } else if (!BACKWARD) {
          float acc = prev[s];
          if (s >= 1) acc = logaddexp(acc, prev[s - 1]);
          if (s >= 2 && ext[s] != (int)blank && ext[s] != ext[s - 2]) acc = logaddexp(acc, prev[s - 2]);
          v = acc + e;
        } else {
          float acc = prev[s];
          if (s + 1 < S) acc = logaddexp(acc, prev[s + 1]);
          if (s + 2 < S && ext[s + 2] != (int)blank && ext[s + 2] != ext[s]) acc = logaddexp(acc, prev[s + 2]);
          v = acc + e;
        }
        now[s] = v;
        ob[t * Smax + s] = v;
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  if (!BACKWARD && tid == 0) {
    const float* last = a[cur ^ 1];
    float ll = last[S - 1];
    if (S > 1) ll = logaddexp(ll, last[S - 2]);
    nll[b] = -ll;
  }
}

template <typename T, bool BACKWARD>
__global__ __launch_bounds__(CTC_THREADS) void ctc_recursion_kernel(
    const T* __restrict__ x, const float* __restrict__ lse, const int64_t* __restrict__ targets,
    const int64_t* __restrict__ in_len, const int64_t* __restrict__ tgt_len, float* __restrict__ out, float* __restrict__ nll,
    int64_t Tmax, int64_t V, int64_t Lmax, int64_t Smax, int64_t blank, const int32_t* __restrict__ rowoff) {
  __shared__ float a[2][CTC_MAX_S + 2];
  __shared__ int ext[CTC_MAX_S];
  __shared__ float em[CTC_CHUNK_FLOATS];
  ctc_recursion_body<T, BACKWARD>(a, ext, em, x, lse, targets, in_len, tgt_len, out, nll, Tmax, V, Lmax, Smax, blank, rowoff);
}
// alpha (blockIdx.y == 0) and beta (blockIdx.y == 1) in one launch: each recursion is a chain of Tmax dependent steps on
// one block per utterance - run back to back they leave the chip idle twice as long
template <typename T>
__global__ __launch_bounds__(CTC_THREADS) void ctc_both_kernel(
    const T* __restrict__ x, const float* __restrict__ lse, const int64_t* __restrict__ targets,
    const int64_t* __restrict__ in_len, const int64_t* __restrict__ tgt_len, float* __restrict__ alpha, float* __restrict__ beta,
    float* __restrict__ nll, int64_t Tmax, int64_t V, int64_t Lmax, int64_t Smax, int64_t blank, const int32_t* __restrict__ rowoff) {
  __shared__ float a[2][CTC_MAX_S + 2];
  __shared__ int ext[CTC_MAX_S];
  __shared__ float em[CTC_CHUNK_FLOATS];
  if (blockIdx.y == 0)
    ctc_recursion_body<T, false>(a, ext, em, x, lse, targets, in_len, tgt_len, alpha, nll, Tmax, V, Lmax, Smax, blank, rowoff);
  else
    ctc_recursion_body<T, true>(a, ext, em, x, lse, targets, in_len, tgt_len, beta, nullptr, Tmax, V, Lmax, Smax, blank, rowoff);
}

// Short targets (2L+1 <= 192 states, i.e. every LS100 / MuST-C batch): the recursion of one utterance runs in ONE wave with
// three states per lane in registers - neighbours come by lane shuffles - while the other three waves of the block gather
// the emissions of the next chunk of time steps into LDS.  The chain of T dependent steps then costs a few hundred cycles
// per step instead of an LDS round trip plus a block barrier.  blockIdx.y: 0 = alpha, 1 = beta.
constexpr int CTCW_S = 192;
constexpr int CTCW_CHUNK = 32;  // time steps staged per chunk: 2 x 32 x 192 floats = 48 KB of LDS

template <typename T>
__global__ __launch_bounds__(256) void ctc_wave_kernel(const T* __restrict__ x, const float* __restrict__ lse,
                                                       const int64_t* __restrict__ targets, const int64_t* __restrict__ in_len,
                                                       const int64_t* __restrict__ tgt_len, float* __restrict__ alpha,
                                                       float* __restrict__ beta, float* __restrict__ nll, int64_t Tmax, int64_t V,
                                                       int64_t Lmax, int64_t Smax, int64_t blank, const int32_t* __restrict__ rowoff) {
  __shared__ float em[2][CTCW_CHUNK * CTCW_S];
  __shared__ int ext[CTCW_S + 2];
  __shared__ float fin[CTCW_S];
  const bool backward = blockIdx.y == 1;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t Tb = min(in_len[b], Tmax);
  const int L = (int)min(tgt_len[b], Lmax);
  const int S = 2 * L + 1;
  float* ob = (backward ? beta : alpha) + (int64_t)b * Tmax * Smax;
  const int64_t r0 = rowoff ? (int64_t)rowoff[b] : (int64_t)b * Tmax;  // first logits / lse row of the utterance
  for (int s2 = tid; s2 < CTCW_S + 2; s2 += 256) ext[s2] = (s2 < S) ? ((s2 & 1) ? (int)targets[b * Lmax + (s2 >> 1)] : (int)blank) : -1;
  for (int64_t i = Tb * Smax + tid; i < Tmax * Smax; i += 256) ob[i] = -INFINITY;  // rows the gradient kernel never reads
  __syncthreads();
  if (Tb <= 0) {
    if (!backward && tid == 0) nll[b] = INFINITY;
    return;
  }
  auto stage = [&](int chunk, int t0, int nthreads) {  // emissions of steps chunk*CHUNK .. in recursion order
    const int64_t c0 = (int64_t)chunk * CTCW_CHUNK;
    const int nt = (int)min((int64_t)CTCW_CHUNK, Tb - c0);
    float* dst = em[chunk & 1];
    for (int i = t0; i < nt * S; i += nthreads) {
      const int tt = i / S, s2 = i - tt * S;
      const int64_t t = backward ? (Tb - 1 - (c0 + tt)) : (c0 + tt);
      const int lab = ext[s2];
      dst[tt * CTCW_S + s2] = (lab >= 0 && lab < V) ? io<T>::ld(x + (r0 + t) * V + lab) - lse[r0 + t] : -INFINITY;
    }
  };
  const int nchunks = (int)((Tb + CTCW_CHUNK - 1) / CTCW_CHUNK);
  stage(0, tid, 256);
  __syncthreads();
  // wave 0: the recursion; lane l owns states 3l, 3l+1, 3l+2
  const int s0 = 3 * lane;
  bool skip[3];  // may the state take the path that jumps over a blank?
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s2 = s0 + j;
    skip[j] = backward ? (s2 + 2 < S && ext[s2 + 2] != (int)blank && ext[s2 + 2] != ext[s2])
                       : (s2 >= 2 && s2 < S && ext[s2] != (int)blank && ext[s2] != ext[s2 - 2]);
  }
  float a0 = -INFINITY, a1 = -INFINITY, a2 = -INFINITY;
  for (int c = 0; c < nchunks; ++c) {
    if (w != 0) {
      if (c + 1 < nchunks) stage(c + 1, tid - 64, 192);
    } else {
      const int64_t c0 = (int64_t)c * CTCW_CHUNK;
      const int nt = (int)min((int64_t)CTCW_CHUNK, Tb - c0);
      const float* e = em[c & 1];
      for (int tt = 0; tt < nt; ++tt) {
        const int64_t step = c0 + tt;
        const int64_t t = backward ? (Tb - 1 - step) : step;
        const float e0 = s0 < S ? e[tt * CTCW_S + s0] : -INFINITY, e1 = s0 + 1 < S ? e[tt * CTCW_S + s0 + 1] : -INFINITY,
                    e2 = s0 + 2 < S ? e[tt * CTCW_S + s0 + 2] : -INFINITY;
        float n0, n1, n2;
        if (step == 0) {
          n0 = (backward ? (s0 >= S - 2 && s0 < S) : (s0 < 2)) ? e0 : -INFINITY;
          n1 = (backward ? (s0 + 1 >= S - 2 && s0 + 1 < S) : (s0 + 1 < 2)) ? e1 : -INFINITY;
          n2 = (backward ? (s0 + 2 >= S - 2 && s0 + 2 < S) : false) ? e2 : -INFINITY;
        } else if (!backward) {
          float p1 = __shfl_up(a1, 1, 64), p2 = __shfl_up(a2, 1, 64);  // states 3l-2, 3l-1
          if (lane == 0) { p1 = -INFINITY; p2 = -INFINITY; }
          float v0 = logaddexp_fast(a0, p2);
          if (skip[0]) v0 = logaddexp_fast(v0, p1);
          float v1 = logaddexp_fast(a1, a0);
          if (skip[1]) v1 = logaddexp_fast(v1, p2);
          float v2 = logaddexp_fast(a2, a1);
          if (skip[2]) v2 = logaddexp_fast(v2, a0);
          n0 = v0 + e0, n1 = v1 + e1, n2 = v2 + e2;
        } else {
          float q0 = __shfl_down(a0, 1, 64), q1 = __shfl_down(a1, 1, 64);  // states 3l+3, 3l+4
          if (lane == 63) { q0 = -INFINITY; q1 = -INFINITY; }
          float v2 = logaddexp_fast(a2, q0);
          if (skip[2]) v2 = logaddexp_fast(v2, q1);
          float v1 = logaddexp_fast(a1, a2);
          if (skip[1]) v1 = logaddexp_fast(v1, q0);
          float v0 = logaddexp_fast(a0, a1);
          if (skip[0]) v0 = logaddexp_fast(v0, a2);
          n0 = v0 + e0, n1 = v1 + e1, n2 = v2 + e2;
        }
        a0 = s0 < S ? n0 : -INFINITY, a1 = s0 + 1 < S ? n1 : -INFINITY, a2 = s0 + 2 < S ? n2 : -INFINITY;
        float* orow = ob + t * Smax;
        if (s0 < S) orow[s0] = a0;
        if (s0 + 1 < S) orow[s0 + 1] = a1;
        if (s0 + 2 < S) orow[s0 + 2] = a2;
      }
    }
    __syncthreads();
  }
  if (!backward) {
    if (w == 0) {
      if (s0 < CTCW_S) fin[s0] = a0;
      if (s0 + 1 < CTCW_S) fin[s0 + 1] = a1;
      if (s0 + 2 < CTCW_S) fin[s0 + 2] = a2;
    }
    __syncthreads();
    if (tid == 0) {
      float ll = fin[S - 1];
      if (S > 1) ll = logaddexp_fast(ll, fin[S - 2]);
      nll[b] = -ll;
    }
  }
}

// loss_b = zero_infinity ? (isinf(nll) ? 0 : nll) : nll
__global__ void ctc_loss_rows_kernel(const float* __restrict__ nll, float* __restrict__ loss_rows, int64_t B, int zero_inf) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < B) {
    const float v = nll[i];
    loss_rows[i] = (zero_inf && isinf(v)) ? 0.f : v;
  }
}

// dx[b,t,v] = gs * ( softmax_t(v) - sum_{s: ext(s)=v} exp(alpha_t(s) + beta_t(s) - lp_t(s) + nll_b) ), 0 for t >= T_b
template <typename T>
__global__ __launch_bounds__(LB) void ctc_grad_kernel(const T* __restrict__ x, const float* __restrict__ lse,
                                                      const float* __restrict__ alpha, const float* __restrict__ beta,
                                                      const float* __restrict__ nll, const int64_t* __restrict__ targets,
                                                      const int64_t* __restrict__ in_len, const int64_t* __restrict__ tgt_len,
                                                      const float* __restrict__ g_dev, float scale, T* __restrict__ dx,
                                                      int64_t Tmax, int64_t V, int64_t Lmax, int64_t Smax, int64_t blank,
                                                      int zero_inf, int ordered, const int32_t* __restrict__ rowoff, int64_t rows_total) {
  extern __shared__ float row[];  // V floats
  __shared__ float occ_s[CTC_MAX_S];  // ordered form: the positions' terms and labels, summed by each label's first occurrence
  __shared__ int lab_s[CTC_MAX_S];
  const int64_t bt = blockIdx.x, b = bt / Tmax, t = bt - b * Tmax;
  const int64_t Tb = min(in_len[b], Tmax);
  if (rowoff) {  // packed rows: the rows behind the last utterance belong to nobody - zeroed by the grid as a whole (block bt takes rows
                 // rowoff[B] + bt, + gridDim.x, ..: the products behind this gradient sum over all rows)
    for (int64_t rz = (int64_t)rowoff[gridDim.x / Tmax] + bt; rz < rows_total; rz += gridDim.x)
      for (int64_t v = threadIdx.x; v < V; v += LB) io<T>::st(dx + rz * V + v, 0.f);
    if (t >= Tb) return;  // positions behind the utterance's length have no row
  }
  const int64_t rw = rowoff ? (int64_t)rowoff[b] + t : bt;  // logits / lse / gradient row; alpha / beta stay [B, Tmax, Smax]
  T* dr = dx + rw * V;
  const float nl = nll[b];
  const bool dead = t >= Tb || (zero_inf && isinf(nl));
  // 16-byte row accesses when the rows allow it (bf16, V % 8 == 0, aligned base)
  const bool vec = sizeof(T) == 2 && (V & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)dx)) & 15) == 0;
  if (dead) {
    if (vec) {
      for (int c = threadIdx.x; c < (int)(V >> 3); c += LB) ((uint4*)dr)[c] = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (int64_t v = threadIdx.x; v < V; v += LB) io<T>::st(dr + v, 0.f);
    }
    return;
  }
  const float gs = scale * (g_dev ? *g_dev : 1.f);
  const T* xr = x + rw * V;
  const float l = lse[rw];
  if (vec) {
    for (int c = threadIdx.x; c < (int)(V >> 3); c += LB) {
      const uint4 q = ((const uint4*)xr)[c];
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        row[8 * c + 2 * i] = __expf(__uint_as_float(w[i] << 16) - l);
        row[8 * c + 2 * i + 1] = __expf(__uint_as_float(w[i] & 0xffff0000u) - l);
      }
    }
  } else {
    for (int64_t v = threadIdx.x; v < V; v += LB) row[v] = __expf(io<T>::ld(xr + v) - l);
  }
  __syncthreads();
  const int L = (int)min(tgt_len[b], Lmax), S = 2 * L + 1;
  const float* ar = alpha + bt * Smax;
  const float* br = beta + bt * Smax;
  if (!ordered) {
    for (int s = threadIdx.x; s < S; s += LB) {
      const int lab = (s & 1) ? (int)targets[b * Lmax + (s >> 1)] : (int)blank;
      if (lab < 0 || lab >= V) continue;
      const float lp = io<T>::ld(xr + lab) - l;
      const float occ = ar[s] + br[s] - lp + nl;
      if (occ > -80.f) atomicAdd(&row[lab], -__expf(occ));
    }
  } else {
    // deterministic form (js2t_set_deterministic): a label that occurs several times in the extended target (every blank, repeated
    // tokens) collects its terms in position order, by the thread of its first occurrence - no float atomics
    for (int s = threadIdx.x; s < S; s += LB) {
      const int lab = (s & 1) ? (int)targets[b * Lmax + (s >> 1)] : (int)blank;
      float term = 0.f;
      if (lab >= 0 && lab < V) {
        const float lp = io<T>::ld(xr + lab) - l;
        const float occ = ar[s] + br[s] - lp + nl;
        if (occ > -80.f) term = -__expf(occ);
      }
      lab_s[s] = (lab >= 0 && lab < V) ? lab : -1;
      occ_s[s] = term;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += LB) {
      const int lab = lab_s[s];
      if (lab < 0) continue;
      bool first = true;
      for (int q = (s & 1); q < s; q += 2)  // same parity only: blanks sit on even positions, labels on odd ones ...
        if (lab_s[q] == lab) { first = false; break; }
      if (first && (s & 1)) {               // ... unless a label equals the blank id
        for (int q = 0; q < s; q += 2)
          if (lab_s[q] == lab) { first = false; break; }
      }
      if (!first) continue;
      float acc = row[lab];
      for (int q = s; q < S; ++q)
        if (lab_s[q] == lab) acc += occ_s[q];
      row[lab] = acc;
    }
  }
  __syncthreads();
  if (vec) {
    for (int c = threadIdx.x; c < (int)(V >> 3); c += LB) {
      uint4 q;
      q.x = pack_bf16x2(gs * row[8 * c + 0], gs * row[8 * c + 1]);
      q.y = pack_bf16x2(gs * row[8 * c + 2], gs * row[8 * c + 3]);
      q.z = pack_bf16x2(gs * row[8 * c + 4], gs * row[8 * c + 5]);
      q.w = pack_bf16x2(gs * row[8 * c + 6], gs * row[8 * c + 7]);
      ((uint4*)dr)[c] = q;
    }
  } else {
    for (int64_t v = threadIdx.x; v < V; v += LB) io<T>::st(dr + v, gs * row[v]);
  }
}

// ---------------------------------------------------------------- CTC best path: collapse repeats, drop blanks
// one wave per utterance: keep[t] = t < len and best[t] != blank and (t == 0 or best[t] != best[t-1]); kept labels are
// compacted with a ballot prefix count, 64 frames per round
__global__ __launch_bounds__(64) void ctc_collapse_kernel(const int64_t* __restrict__ best, const int64_t* __restrict__ in_len,
                                                         int64_t* __restrict__ out_ids, int64_t* __restrict__ out_len, int64_t T,
                                                         int64_t blank, int64_t pad) {
  const int64_t b = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t len = min(in_len[b], T);
  const int64_t* row = best + b * T;
  int64_t* orow = out_ids + b * T;
  int64_t n = 0;
  for (int64_t t0 = 0; t0 < T; t0 += 64) {
    const int64_t t = t0 + lane;
    const int64_t cur = t < len ? row[t] : blank;
    const int64_t prev = (t > 0 && t < len) ? row[t - 1] : -1;
    const bool keep = t < len && cur != blank && cur != prev;
    const unsigned long long m = __ballot(keep);
    if (keep) orow[n + __popcll(m & ((1ull << lane) - 1ull))] = cur;
    n += __popcll(m);
  }
  for (int64_t t = n + lane; t < T; t += 64) orow[t] = pad;
  if (lane == 0) out_len[b] = n;
}

}  // namespace

#define DISPATCH_DT(dt, T, ...)                                  \
  do {                                                           \
    if ((dt) == JS2T_F32) { typedef float T; __VA_ARGS__; }      \
    else if ((dt) == JS2T_BF16) { typedef uint16_t T; __VA_ARGS__; } \
    else { js2t_set_error("bad dtype %d", (int)(dt)); return JS2T_ERR_INVALID; } \
  } while (0)

extern "C" int js2t_row_lse(const void* x, float* lse, int64_t* argmax, int64_t rows, int64_t V, int dt, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(x && lse && rows > 0 && V > 0, "row_lse: bad arguments");
  if (dt == JS2T_BF16 && (V & 7) == 0 && V <= 8 * LB * RL_CH && (((uintptr_t)x) & 15) == 0) {
    hipLaunchKernelGGL(row_lse_vec_kernel, dim3((unsigned)rows), dim3(LB), 0, (hipStream_t)stream, (const uint16_t*)x, lse, argmax, rows, V);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((row_lse_kernel<T>), dim3((unsigned)rows), dim3(LB), 0, (hipStream_t)stream,
                                        (const T*)x, lse, argmax, rows, V));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_ctc_collapse(const int64_t* best, const int64_t* in_len, int64_t* out_ids, int64_t* out_len, int64_t B, int64_t T,
                                 int64_t blank, int64_t pad, js2t_stream stream) {
  if (B == 0) return JS2T_OK;
  JS2T_CHECK(best && in_len && out_ids && out_len && B > 0 && T > 0, "ctc_collapse: bad arguments");
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, best, in_len, out_ids, out_len, T, blank,
                     pad);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_log_softmax(const void* x, int dt, void* y, int y_dt, int64_t rows, int64_t V, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(x && y && rows > 0 && V > 0, "log_softmax: bad arguments");
  DISPATCH_DT(dt, T, DISPATCH_DT(y_dt, TO, hipLaunchKernelGGL((log_softmax_kernel<T, TO>), dim3((unsigned)rows), dim3(LB), 0,
                                                              (hipStream_t)stream, (const T*)x, (TO*)y, rows, V)));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_xent_fwd(const void* logits, int dt, const int64_t* trg, float* loss_rows, float* correct_rows, float* lse,
                             int64_t rows, int64_t V, int64_t pad_idx, float smoothing, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(logits && trg && loss_rows && correct_rows && lse && V > 2, "xent_fwd: bad arguments");
  if (dt == JS2T_F32 && (V & 3) == 0 && V < (int64_t(1) << 31) && (((uintptr_t)logits) & 15) == 0) {
    hipLaunchKernelGGL(xent_fwd_wave_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const float*)logits, trg,
                       loss_rows, correct_rows, lse, rows, V, pad_idx, smoothing);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((xent_fwd_kernel<T>), dim3((unsigned)rows), dim3(LB), 0, (hipStream_t)stream,
                                        (const T*)logits, trg, loss_rows, correct_rows, lse, rows, V, pad_idx, smoothing));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_xent_bwd(const void* logits, int dt, const int64_t* trg, const float* lse, const float* g_dev, float scale,
                             void* dlogits, int64_t rows, int64_t V, int64_t pad_idx, float smoothing, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(logits && trg && lse && dlogits && V > 2, "xent_bwd: bad arguments");
  if (dt == JS2T_F32 && (V & 7) == 0 && V < (int64_t(1) << 31) && rows < (int64_t(1) << 31) && rows * (V >> 3) < (int64_t(1) << 31) * 256 &&
      ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) == 0) {
    hipLaunchKernelGGL((xent_bwd_flat_kernel<float>), dim3((unsigned)cdiv(rows * (V >> 3), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)logits, trg, lse, g_dev, scale, (float*)dlogits, (int)rows, (int)(V >> 3), V, pad_idx, smoothing);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((xent_bwd_kernel<T>), dim3((unsigned)rows), dim3(LB), 0, (hipStream_t)stream,
                                        (const T*)logits, trg, lse, g_dev, scale, (T*)dlogits, rows, V, pad_idx, smoothing));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_xent_bwd_as(const void* logits, int dt, const int64_t* trg, const float* lse, const float* g_dev, float scale,
                                void* dlogits, int out_dt, int64_t rows, int64_t V, int64_t pad_idx, float smoothing, js2t_stream stream) {
  if (out_dt == dt) return js2t_xent_bwd(logits, dt, trg, lse, g_dev, scale, dlogits, rows, V, pad_idx, smoothing, stream);
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(logits && trg && lse && dlogits && V > 2, "xent_bwd_as: bad arguments");
  JS2T_CHECK(dt == JS2T_F32 && out_dt == JS2T_BF16, "xent_bwd_as: f32 logits -> bf16 gradient, or equal types");
  if ((V & 7) == 0 && V < (int64_t(1) << 31) && rows < (int64_t(1) << 31) && rows * (V >> 3) < (int64_t(1) << 31) * 256 &&
      ((((uintptr_t)logits) | ((uintptr_t)dlogits)) & 15) == 0) {
    hipLaunchKernelGGL((xent_bwd_flat_kernel<uint16_t>), dim3((unsigned)cdiv(rows * (V >> 3), 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)logits, trg, lse, g_dev, scale, (uint16_t*)dlogits, (int)rows, (int)(V >> 3), V, pad_idx, smoothing);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  hipLaunchKernelGGL((xent_bwd_kernel<float, uint16_t>), dim3((unsigned)rows), dim3(LB), 0, (hipStream_t)stream, (const float*)logits, trg,
                     lse, g_dev, scale, (uint16_t*)dlogits, rows, V, pad_idx, smoothing);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// running statistics of the train step in one launch (the reference adds six Python scalars, training.py:566-586)
__global__ void train_stats_kernel(double* __restrict__ stats, const float* total, const float* nll, const float* ctc,
                                   const int64_t* n_correct, double inv_norm, double nseqs, double ntokens, float* norm_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double t = (double)total[0] * inv_norm;
  stats[0] += t;
  if (nll) stats[1] += (double)nll[0] * inv_norm;
  if (ctc) stats[2] += (double)ctc[0] * inv_norm;
  if (n_correct) stats[3] += (double)n_correct[0];
  stats[4] += nseqs;
  stats[5] += ntokens;
  if (norm_out) norm_out[0] = (float)t;
}
extern "C" int js2t_train_stats(double* stats6, const float* total, const float* nll, const float* ctc, const int64_t* n_correct,
                                double inv_norm, double nseqs, double ntokens, float* norm_out, js2t_stream stream) {
  JS2T_CHECK(stats6 && total, "train_stats: null pointer");
  hipLaunchKernelGGL(train_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats6, total, nll, ctc, n_correct, inv_norm, nseqs,
                     ntokens, norm_out);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_sum_f32(const float* x, int64_t n, float* out, js2t_stream stream) {
  JS2T_CHECK(x && out && n >= 0, "sum_f32: bad arguments");
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, out);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_ctc_alpha(const void* logits, int dt, const float* lse, const int64_t* targets, const int64_t* in_len,
                              const int64_t* tgt_len, float* alpha, float* beta, float* nll, float* loss_rows, int64_t B,
                              int64_t T_, int64_t V, int64_t Lmax, int64_t blank, int zero_infinity, const int32_t* row_offsets,
                              js2t_stream stream) {
  if (B == 0) return JS2T_OK;
  JS2T_CHECK(logits && lse && targets && in_len && tgt_len && alpha && nll && loss_rows, "ctc_alpha: null pointer");
  JS2T_CHECK(2 * Lmax + 1 <= CTC_MAX_S, "ctc_alpha: target length %lld exceeds %d", (long long)Lmax, (CTC_MAX_S - 1) / 2);
  const int64_t Smax = 2 * Lmax + 1;
  hipStream_t s = (hipStream_t)stream;
  if (Smax <= CTCW_S) {  // short targets: register-resident recursion, alpha (and beta when asked for) in one launch
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((ctc_wave_kernel<T>), dim3((unsigned)B, beta ? 2 : 1), dim3(256), 0, s, (const T*)logits, lse,
                                          targets, in_len, tgt_len, alpha, beta, nll, T_, V, Lmax, Smax, blank, row_offsets));
  } else if (beta) {
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((ctc_both_kernel<T>), dim3((unsigned)B, 2), dim3(CTC_THREADS), 0, s, (const T*)logits, lse,
                                          targets, in_len, tgt_len, alpha, beta, nll, T_, V, Lmax, Smax, blank, row_offsets));
  } else {
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((ctc_recursion_kernel<T, false>), dim3((unsigned)B), dim3(CTC_THREADS), 0, s,
                                          (const T*)logits, lse, targets, in_len, tgt_len, alpha, nll, T_, V, Lmax, Smax, blank, row_offsets));
  }
  JS2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctc_loss_rows_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, nll, loss_rows, B, zero_infinity);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_ctc_bwd(const void* logits, int dt, const float* lse, const int64_t* targets, const int64_t* in_len,
                            const int64_t* tgt_len, const float* alpha, float* beta, const float* nll, const float* g_dev,
                            float scale, void* dlogits, int64_t B, int64_t T_, int64_t V, int64_t Lmax, int64_t blank,
                            int zero_infinity, int beta_ready, const int32_t* row_offsets, int64_t packed_rows, js2t_stream stream) {
  if (B == 0) return JS2T_OK;
  JS2T_CHECK(logits && lse && targets && in_len && tgt_len && alpha && beta && nll && dlogits, "ctc_bwd: null pointer");
  JS2T_CHECK(2 * Lmax + 1 <= CTC_MAX_S, "ctc_bwd: target length %lld exceeds %d", (long long)Lmax, (CTC_MAX_S - 1) / 2);
  JS2T_CHECK(V * 4 <= 160 * 1024 - 10 * 1024, "ctc_bwd: vocabulary %lld too large for the LDS-staged gradient row", (long long)V);
  const int64_t Smax = 2 * Lmax + 1;
  hipStream_t s = (hipStream_t)stream;
  if (!beta_ready) {
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((ctc_recursion_kernel<T, true>), dim3((unsigned)B), dim3(CTC_THREADS), 0, s,
                                          (const T*)logits, lse, targets, in_len, tgt_len, beta, (float*)nullptr, T_, V, Lmax,
                                          Smax, blank, row_offsets));
    JS2T_LAUNCH_CHECK();
  }
  const size_t lds = (size_t)V * sizeof(float);
  if (dt == JS2T_F32) {
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)ctc_grad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ctc_grad_kernel<float>), dim3((unsigned)(B * T_)), dim3(LB), lds, s, (const float*)logits, lse, alpha,
                       beta, nll, targets, in_len, tgt_len, g_dev, scale, (float*)dlogits, T_, V, Lmax, Smax, blank,
                       zero_infinity, g_js2t_deterministic, row_offsets, packed_rows);
  } else {
    if (lds > 48 * 1024) hipFuncSetAttribute((const void*)ctc_grad_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((ctc_grad_kernel<uint16_t>), dim3((unsigned)(B * T_)), dim3(LB), lds, s, (const uint16_t*)logits, lse,
                       alpha, beta, nll, targets, in_len, tgt_len, g_dev, scale, (uint16_t*)dlogits, T_, V, Lmax, Smax, blank,
                       zero_infinity, g_js2t_deterministic, row_offsets, packed_rows);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
