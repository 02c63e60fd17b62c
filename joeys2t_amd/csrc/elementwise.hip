// HBM-bound element-wise and data-movement kernels of the path (casts, GLU, positional encoding +
// dropout, dropout / activation backward, embedding gather / scatter, column sums, Conv1d weight repack,
// col2im gather, subsampled lengths + padding mask).  All are grid-stride, 16-byte-per-lane where the
// layout allows; math in f32.
#include "common.hpp"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_grid(int64_t work_items) {
  int64_t g = (work_items + EW_THREADS - 1) / EW_THREADS;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;  // 256 CUs x 16 blocks, grid-stride beyond
  return (int)g;
}

// ---- vector-of-4 accessors (4 consecutive elements; caller guarantees 4-element alignment) ----------
template <typename T> struct vec4;
template <> struct vec4<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const float4 x = *(const float4*)p; v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct vec4<uint16_t> {
  static __device__ __forceinline__ void ld(const uint16_t* p, float (&v)[4]) {
    const uint2 x = *(const uint2*)p;
    v[0] = __uint_as_float(x.x << 16); v[1] = __uint_as_float(x.x & 0xffff0000u);
    v[2] = __uint_as_float(x.y << 16); v[3] = __uint_as_float(x.y & 0xffff0000u);
  }
  static __device__ __forceinline__ void st(uint16_t* p, const float (&v)[4]) {
    uint2 x;
    x.x = pack_bf16x2(v[0], v[1]);
    x.y = pack_bf16x2(v[2], v[3]);
    *(uint2*)p = x;
  }
};

// ---------------------------------------------------------------- cast
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float v[4];
    vec4<S>::ld(src + 4 * i, v);
    vec4<D>::st(dst + 4 * i, v);
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    io<D>::st(dst + i, io<S>::ld(src + i));
}

template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, float a, const T* __restrict__ y, float b, T* __restrict__ out,
                             int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = a * io<T>::ld(x + i);
    if (y) v += b * io<T>::ld(y + i);
    io<T>::st(out + i, v);
  }
}

// ---------------------------------------------------------------- GLU
// valid_t (device scalar, optional) with rows = batch x T_: positions t >= *valid_t of every batch entry are written as 0 /
// get a zero gradient - the tensor then looks to the next convolution as if it had been cropped to *valid_t positions
// (js2t_glu_fwd_crop)
template <typename T>
__global__ void glu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int64_t C, int64_t T_,
                               const int64_t* __restrict__ valid_t) {
  const int64_t total = rows * C, vt = valid_t ? *valid_t : T_;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C, c = i - r * C;
    const float a = io<T>::ld(x + r * 2 * C + c), b = io<T>::ld(x + r * 2 * C + C + c);
    io<T>::st(y + i, (r % T_) < vt ? a / (1.f + __expf(-b)) : 0.f);
  }
}
template <typename T>
__global__ void glu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t rows,
                               int64_t C, int64_t T_, const int64_t* __restrict__ valid_t) {
  const int64_t total = rows * C, vt = valid_t ? *valid_t : T_;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C, c = i - r * C;
    const float a = io<T>::ld(x + r * 2 * C + c), b = io<T>::ld(x + r * 2 * C + C + c);
    const float g = (r % T_) < vt ? io<T>::ld(dy + i) : 0.f;
    const float s = 1.f / (1.f + __expf(-b));
    io<T>::st(dx + r * 2 * C + c, g * s);
    io<T>::st(dx + r * 2 * C + C + c, g * a * s * (1.f - s));
  }
}

// ---------------------------------------------------------------- y = dropout(x + pe (+ extra))
template <typename T>
__global__ void add_pe_dropout_kernel(const T* __restrict__ x, const float* __restrict__ pe, const T* __restrict__ extra,
                                      T* __restrict__ y, int64_t B, int64_t T_, int64_t D, float p,
                                      const uint64_t* rng, uint32_t stream) {
  // one thread per 4 columns (D % 4 == 0 enforced by the host)
  const int64_t D4 = D >> 2, total = B * T_ * D4;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t dkey = p > 0.f ? dropout_key(rng, stream) : 0u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D4, c4 = i - row * D4, t = row % T_;
    float v[4], e[4];
    vec4<T>::ld(x + row * D + 4 * c4, v);
    if (pe) {  // pe == nullptr: plain dropout
      const float4 pv = *(const float4*)(pe + t * D + 4 * c4);
      v[0] += pv.x; v[1] += pv.y; v[2] += pv.z; v[3] += pv.w;
    }
    if (extra) {
      vec4<T>::ld(extra + row * D + 4 * c4, e);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += e[k];
    }
    if (p > 0.f) {
      const uint32_t keep = dropout_keep4_key(dkey, (uint32_t)row, (uint32_t)c4, p);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ((keep >> k) & 1u) ? v[k] * sc : 0.f;
    }
    vec4<T>::st(y + row * D + 4 * c4, v);
  }
}

template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t rows, int64_t cols, float p,
                                   const uint64_t* rng, uint32_t stream) {
  const int64_t c4n = (cols + 3) >> 2, total = rows * c4n;
  const float sc = 1.f / (1.f - p);
  const uint32_t dkey = dropout_key(rng, stream);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4n, c4 = i - row * c4n;
    const uint32_t keep = dropout_keep4_key(dkey, (uint32_t)row, (uint32_t)c4, p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = 4 * c4 + k;
      if (c < cols) {
        const float g = io<T>::ld(dy + row * cols + c);
        io<T>::st(dx + row * cols + c, ((keep >> k) & 1u) ? g * sc : 0.f);
      }
    }
  }
}

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ z, T* __restrict__ dz, int64_t n, int act,
                               float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    io<T>::st(dz + i, io<T>::ld(dh + i) * act_grad(io<T>::ld(z + i), act) * scale);
}

// ---------------------------------------------------------------- embedding
template <typename TT, typename TO>
__global__ void embed_fwd_kernel(const int64_t* __restrict__ ids, const TT* __restrict__ table, TO* __restrict__ out,
                                 int64_t n_ids, int64_t D, int64_t vocab, float scale) {
  const int64_t total = n_ids * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D, c = i - r * D;
    const int64_t id = ids[r];
    const float v = (id >= 0 && id < vocab) ? io<TT>::ld(table + id * D + c) * scale : 0.f;
    io<TO>::st(out + i, v);
  }
}
template <typename T>
__global__ void embed_bwd_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout, float* __restrict__ dtable,
                                 int64_t n_ids, int64_t D, int64_t vocab, float scale, int64_t pad_idx) {
  const int64_t total = n_ids * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D, c = i - r * D;
    const int64_t id = ids[r];
    if (id < 0 || id >= vocab || id == pad_idx) continue;
    atomicAdd(dtable + id * D + c, scale * io<T>::ld(dout + i));
  }
}

// deterministic form: block r owns table row ids[r] iff no earlier position holds the same id, and then adds the rows of
// all positions with that id in position order (n_ids is a few thousand: the scans are noise)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_ordered_kernel(const int64_t* __restrict__ ids, const T* __restrict__ dout,
                                                                float* __restrict__ dtable, int64_t n_ids, int64_t D, int64_t vocab,
                                                                float scale, int64_t pad_idx) {
  __shared__ int earlier;
  const int64_t r = blockIdx.x, id = ids[r];
  if (id < 0 || id >= vocab || id == pad_idx) return;
  if (threadIdx.x == 0) earlier = 0;
  __syncthreads();
  for (int64_t q = threadIdx.x; q < r; q += blockDim.x)
    if (ids[q] == id) earlier = 1;  // (benign race: every writer stores 1)
  __syncthreads();
  if (earlier) return;
  for (int64_t c = threadIdx.x; c < D; c += blockDim.x) {
    float acc = dtable[id * D + c];
    for (int64_t q = r; q < n_ids; ++q)
      if (ids[q] == id) acc += scale * io<T>::ld(dout + q * D + c);
    dtable[id * D + c] = acc;
  }
}

// ---------------------------------------------------------------- column sums (two-stage, deterministic)
// Stage 1: block = 64 column-chunks x 4 row-lanes over a 64-row slab; a thread owns 8 consecutive columns (16-byte
// bf16 loads) when the row length allows, else one column.  Stage 2 adds the per-slab partials in a fixed order with
// 4 partial-lanes per column.
constexpr int CS_ROWS_PER_BLOCK = 64;
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ partial, int64_t rows,
                                                             int64_t cols) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS_PER_BLOCK, r1 = min(rows, r0 + CS_ROWS_PER_BLOCK);
  float s = 0.f;
  if (c < cols)
    for (int64_t r = r0 + sub; r < r1; r += 4) s += io<T>::ld(x + r * cols + c);
  sh[sub][cl] = s;
  __syncthreads();
  if (sub == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// 8 columns per thread (cols % 8 == 0, 16-byte aligned rows): block covers 512 columns
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const uint16_t* __restrict__ x, float* __restrict__ partial,
                                                                 int64_t rows, int64_t cols) {
  __shared__ float sh[4][64][9];
  const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t c = ((int64_t)blockIdx.x * 64 + cl) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS_PER_BLOCK, r1 = min(rows, r0 + CS_ROWS_PER_BLOCK);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cols)
    for (int64_t r = r0 + sub; r < r1; r += 4) {
      const uint4 v = *(const uint4*)(x + r * cols + c);
      const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s[2 * i] += __uint_as_float(vv[i] << 16);
        s[2 * i + 1] += __uint_as_float(vv[i] & 0xffff0000u);
      }
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) sh[sub][cl][i] = s[i];
  __syncthreads();
  if (sub == 0 && c < cols) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      partial[(int64_t)blockIdx.y * cols + c + i] = (sh[0][cl][i] + sh[1][cl][i]) + (sh[2][cl][i] + sh[3][cl][i]);
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int64_t nparts,
                                                           int64_t cols, int accumulate) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int64_t p = sub;
    for (; p + 4 < nparts; p += 8) {
      s0 += partial[p * cols + c];
      s1 += partial[(p + 4) * cols + c];
    }
    for (; p < nparts; p += 4) s0 += partial[p * cols + c];
  }
  sh[sub][cl] = s0 + s1;
  __syncthreads();
  if (sub == 0 && c < cols) {
    const float tot = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
    out[c] = accumulate ? out[c] + tot : tot;
  }
}

// ---------------------------------------------------------------- conv weight repack
template <typename D>
__global__ void conv_pack_kernel(const float* __restrict__ w, D* __restrict__ wp, int64_t cout, int64_t cin, int64_t k) {
  const int64_t total = cout * cin * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // destination index i = (o, kw, c)
    const int64_t o = i / (k * cin), rem = i - o * k * cin, kw = rem / cin, c = rem - kw * cin;
    io<D>::st(wp + i, w[(o * cin + c) * k + kw]);
  }
}
__global__ void conv_unpack_grad_kernel(const float* __restrict__ dwp_t, float* __restrict__ dw, int64_t cout, int64_t cin,
                                        int64_t k, int accumulate) {
  const int64_t total = cout * cin * k;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // destination index i = (o, c, kw) in torch layout; source [(kw*cin + c), o]
    const int64_t o = i / (cin * k), rem = i - o * cin * k, c = rem / k, kw = rem - c * k;
    const float v = dwp_t[(kw * cin + c) * cout + o];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int64_t B, int64_t tin, int64_t tout,
                              int64_t C, int64_t K, int64_t stride, int64_t pad) {
  const int64_t total = B * tin * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (tin * C), rem = i - b * tin * C, tau = rem / C, c = rem - tau * C;
    float s = 0.f;
    for (int64_t kw = 0; kw < K; ++kw) {
      const int64_t num = tau + pad - kw;
      if (num < 0 || num % stride) continue;
      const int64_t t = num / stride;
      if (t >= tout) continue;
      s += io<T>::ld(dcol + (b * tout + t) * (K * C) + kw * C + c);
    }
    io<T>::st(dx + i, s);
  }
}

// col[(b*tout + t), kw*C + c] = x[b, t*stride - pad + kw, c] (0 outside the utterance): the A operand of the strided
// convolution as a plain row-major matrix, 16 bytes per thread (C % 8 == 0 for bf16, C % 4 == 0 for f32)
template <typename T, int VEC>
__global__ void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int64_t B, int64_t tin, int64_t tout, int64_t C, int64_t K,
                              int64_t stride, int64_t pad) {
  const int64_t cv = C / VEC, total = B * tout * K * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / (K * cv), rem = i - row * (K * cv), kw = rem / cv, c = (rem - kw * cv) * VEC;
    const int64_t b = row / tout, t = row - b * tout, tau = t * stride - pad + kw;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (tau >= 0 && tau < tin) v = *(const uint4*)(x + (b * tin + tau) * C + c);
    *(uint4*)(col + row * (K * C) + kw * C + c) = v;
  }
}

// bf16, C % 8 == 0: one 16-byte piece of dx per thread, the (at most ceil(K / stride)) contributing taps summed in f32
__global__ void col2im_vec_kernel(const uint16_t* __restrict__ dcol, uint16_t* __restrict__ dx, int64_t B, int64_t tin, int64_t tout,
                                  int64_t C, int64_t K, int64_t stride, int64_t pad) {
  const int64_t cv = C >> 3, total = B * tin * cv;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / (tin * cv), rem = i - b * tin * cv, tau = rem / cv, c = (rem - tau * cv) << 3;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int64_t kw = 0; kw < K; ++kw) {
      const int64_t num = tau + pad - kw;
      if (num < 0 || num % stride) continue;
      const int64_t t = num / stride;
      if (t >= tout) continue;
      const uint4 q = *(const uint4*)(dcol + (b * tout + t) * (K * C) + kw * C + c);
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[2 * j] += __uint_as_float(w[j] << 16);
        s[2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
      }
    }
    uint4 o;
    o.x = pack_bf16x2(s[0], s[1]);
    o.y = pack_bf16x2(s[2], s[3]);
    o.z = pack_bf16x2(s[4], s[5]);
    o.w = pack_bf16x2(s[6], s[7]);
    *(uint4*)(dx + (b * tin + tau) * C + c) = o;
  }
}

__global__ void subsample_len_mask_kernel(const int64_t* __restrict__ lengths, int64_t* __restrict__ out_lengths,
                                          uint8_t* __restrict__ mask, int64_t B, int64_t T_out, int k0, int k1, int k2, int k3,
                                          int n_layers) {
  const int ks[4] = {k0, k1, k2, k3};
  const int64_t total = B * T_out;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / T_out, t = i - b * T_out;
    int64_t len = lengths[b];
    for (int l = 0; l < n_layers; ++l) {
      // ((len + 2*(k//2) - (k-1) - 1) / 2 + 1).floor() evaluated in float like the reference
      const float f = ((float)len + 2.f * (float)(ks[l] / 2) - (float)(ks[l] - 1) - 1.f) / 2.f + 1.f;
      len = (int64_t)floorf(f);
    }
    if (t == 0) out_lengths[b] = len;
    mask[i] = t < len ? 1 : 0;
  }
}

// Transposed bf16 shadows of 2-D parameter groups: table[g] = {element offset, rows, cols, first tile}; group g's
// [rows, cols] block at src + offset is written as [cols, rows] at dst + offset.  One 64x64 tile per block through LDS
// (coalesced reads along cols, coalesced writes along rows).
__global__ __launch_bounds__(256) void transpose_groups_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                               const int64_t* __restrict__ table, int n_groups) {
  __shared__ uint16_t tile[64][66];
  const int64_t bid = blockIdx.x;
  int lo = 0, hi = n_groups - 1;
  while (lo < hi) {  // last group whose first tile <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 4 + 3] <= bid) lo = mid; else hi = mid - 1;
  }
  const int64_t off = table[lo * 4], R = table[lo * 4 + 1], Cc = table[lo * 4 + 2], t0 = table[lo * 4 + 3];
  const int64_t tiles_c = (Cc + 63) >> 6;
  const int64_t tr = (bid - t0) / tiles_c, tc = (bid - t0) - tr * tiles_c;
  if (((R | Cc | off) & 7) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
    // 16-byte accesses on both sides (all weight matrices of the models here): rows in as 8-element pieces, columns
    // out as 8-element pieces gathered from the padded tile (33-dword row pitch: the 8 x 8 lanes of a wave hit 32
    // different banks, the two halves of a dword are read by neighbouring lanes)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int o = threadIdx.x + 256 * k, r = o >> 3, c8 = (o & 7) << 3;
      const int64_t rr = tr * 64 + r, cc = tc * 64 + c8;
      if (rr < R && cc < Cc) {
        const uint4 q = *(const uint4*)(src + off + rr * Cc + cc);
        uint32_t* d = (uint32_t*)&tile[r][c8];
        d[0] = q.x, d[1] = q.y, d[2] = q.z, d[3] = q.w;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int o = threadIdx.x + 256 * k, c = o >> 3, r8 = (o & 7) << 3;
      const int64_t cc = tc * 64 + c, rr = tr * 64 + r8;
      if (rr < R && cc < Cc) {
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[r8 + 2 * j][c] | ((uint32_t)tile[r8 + 2 * j + 1][c] << 16);
        *(uint4*)(dst + off + cc * R + rr) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int64_t rr = tr * 64 + r, cc = tc * 64 + tx;
    if (rr < R && cc < Cc) tile[r][tx] = src[off + rr * Cc + cc];
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int64_t cc = tc * 64 + c, rr = tr * 64 + tx;
    if (rr < R && cc < Cc) dst[off + cc * R + rr] = tile[tx][c];
  }
}

}  // namespace

#define DISPATCH_DT(dt, T, ...)                                  \
  do {                                                           \
    if ((dt) == JS2T_F32) { typedef float T; __VA_ARGS__; }      \
    else if ((dt) == JS2T_BF16) { typedef uint16_t T; __VA_ARGS__; } \
    else { js2t_set_error("bad dtype %d", (int)(dt)); return JS2T_ERR_INVALID; } \
  } while (0)

extern "C" int js2t_cast(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, js2t_stream stream) {
  if (n == 0) return JS2T_OK;
  JS2T_CHECK(src && dst && n > 0, "cast: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int g = ew_grid(n / 4 + 1);
  DISPATCH_DT(src_dt, S, DISPATCH_DT(dst_dt, D, hipLaunchKernelGGL((cast_kernel<S, D>), dim3(g), dim3(EW_THREADS), 0, s,
                                                                    (const S*)src, (D*)dst, n)));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_axpby(const void* x, float a, const void* y, float b, void* out, int64_t n, int dt, js2t_stream stream) {
  if (n == 0) return JS2T_OK;
  JS2T_CHECK(x && out && n > 0, "axpby: bad arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((axpby_kernel<T>), dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                                        (const T*)x, a, (const T*)y, b, (T*)out, n));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_glu_fwd_crop(const void* x, void* y, int64_t rows, int64_t C, int64_t T_, const int64_t* valid_t, int dt,
                                 js2t_stream stream) {
  if (rows * C == 0) return JS2T_OK;
  JS2T_CHECK(x && y && rows > 0 && C > 0 && T_ > 0 && rows % T_ == 0, "glu_fwd: bad arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((glu_fwd_kernel<T>), dim3(ew_grid(rows * C)), dim3(EW_THREADS), 0,
                                        (hipStream_t)stream, (const T*)x, (T*)y, rows, C, T_, valid_t));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
extern "C" int js2t_glu_fwd(const void* x, void* y, int64_t rows, int64_t C, int dt, js2t_stream stream) {
  return js2t_glu_fwd_crop(x, y, rows, C, rows > 0 ? rows : 1, nullptr, dt, stream);
}
extern "C" int js2t_glu_bwd_crop(const void* x, const void* dy, void* dx, int64_t rows, int64_t C, int64_t T_, const int64_t* valid_t,
                                 int dt, js2t_stream stream) {
  if (rows * C == 0) return JS2T_OK;
  JS2T_CHECK(x && dy && dx && rows > 0 && C > 0 && T_ > 0 && rows % T_ == 0, "glu_bwd: bad arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((glu_bwd_kernel<T>), dim3(ew_grid(rows * C)), dim3(EW_THREADS), 0,
                                        (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, rows, C, T_, valid_t));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
extern "C" int js2t_glu_bwd(const void* x, const void* dy, void* dx, int64_t rows, int64_t C, int dt, js2t_stream stream) {
  return js2t_glu_bwd_crop(x, dy, dx, rows, C, rows > 0 ? rows : 1, nullptr, dt, stream);
}

extern "C" int js2t_add_pe_dropout(const void* x, const float* pe, const void* extra, void* y, int64_t B, int64_t T_,
                                   int64_t D, int dt, float p, const uint64_t* rng_state, uint32_t rng_stream,
                                   js2t_stream stream) {
  if (B * T_ * D == 0) return JS2T_OK;
  JS2T_CHECK(x && y, "add_pe_dropout: null pointer");
  JS2T_CHECK(D % 4 == 0, "add_pe_dropout: D must be a multiple of 4 (got %lld)", (long long)D);
  JS2T_CHECK(p >= 0.f && p < 1.f && (p == 0.f || rng_state), "add_pe_dropout: bad dropout arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((add_pe_dropout_kernel<T>), dim3(ew_grid(B * T_ * D / 4)), dim3(EW_THREADS), 0,
                                        (hipStream_t)stream, (const T*)x, pe, (const T*)extra, (T*)y, B, T_, D, p,
                                        rng_state, rng_stream));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_dropout_bwd(const void* dy, void* dx, int64_t rows, int64_t cols, int dt, float p,
                                const uint64_t* rng_state, uint32_t rng_stream, js2t_stream stream) {
  if (rows * cols == 0) return JS2T_OK;
  JS2T_CHECK(dy && dx && rng_state && p > 0.f && p < 1.f, "dropout_bwd: bad arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((dropout_bwd_kernel<T>), dim3(ew_grid(rows * ((cols + 3) / 4))), dim3(EW_THREADS),
                                        0, (hipStream_t)stream, (const T*)dy, (T*)dx, rows, cols, p, rng_state, rng_stream));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_act_bwd(const void* dh, const void* z, void* dz, int64_t n, int act, int dt, float scale,
                            js2t_stream stream) {
  if (n == 0) return JS2T_OK;
  JS2T_CHECK(dh && z && dz && n > 0, "act_bwd: bad arguments");
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                                        (const T*)dh, (const T*)z, (T*)dz, n, act, scale));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_embed_fwd(const int64_t* ids, const void* table, int table_dt, void* out, int out_dt, int64_t n_ids,
                              int64_t D, int64_t vocab, float scale, js2t_stream stream) {
  if (n_ids * D == 0) return JS2T_OK;
  JS2T_CHECK(ids && table && out, "embed_fwd: null pointer");
  DISPATCH_DT(table_dt, TT, DISPATCH_DT(out_dt, TO, hipLaunchKernelGGL((embed_fwd_kernel<TT, TO>), dim3(ew_grid(n_ids * D)),
                                                                       dim3(EW_THREADS), 0, (hipStream_t)stream, ids,
                                                                       (const TT*)table, (TO*)out, n_ids, D, vocab, scale)));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
extern "C" int js2t_embed_bwd(const int64_t* ids, const void* dout, int dout_dt, float* dtable, int64_t n_ids, int64_t D,
                              int64_t vocab, float scale, int64_t pad_idx, js2t_stream stream) {
  if (n_ids * D == 0) return JS2T_OK;
  JS2T_CHECK(ids && dout && dtable, "embed_bwd: null pointer");
  if (g_js2t_deterministic) {
    DISPATCH_DT(dout_dt, T, hipLaunchKernelGGL((embed_bwd_ordered_kernel<T>), dim3((unsigned)n_ids), dim3(256), 0, (hipStream_t)stream, ids,
                                               (const T*)dout, dtable, n_ids, D, vocab, scale, pad_idx));
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  DISPATCH_DT(dout_dt, T, hipLaunchKernelGGL((embed_bwd_kernel<T>), dim3(ew_grid(n_ids * D)), dim3(EW_THREADS), 0,
                                             (hipStream_t)stream, ids, (const T*)dout, dtable, n_ids, D, vocab, scale, pad_idx));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int64_t js2t_colsum_partial_rows(int64_t rows) { return (rows + CS_ROWS_PER_BLOCK - 1) / CS_ROWS_PER_BLOCK; }

extern "C" int js2t_colsum(const void* x, int dt, float* out, float* partial, int64_t rows, int64_t cols, int accumulate,
                           js2t_stream stream) {
  if (cols == 0) return JS2T_OK;
  if (rows == 0) {  // empty sum: zero (or leave the accumulator untouched)
    JS2T_CHECK(out, "colsum: bad arguments");
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * cols, (hipStream_t)stream);
      JS2T_CHECK(e == hipSuccess, "colsum: memset failed: %s", hipGetErrorString(e));
    }
    return JS2T_OK;
  }
  JS2T_CHECK(x && out && partial && rows > 0, "colsum: bad arguments");
  const int64_t nparts = js2t_colsum_partial_rows(rows);
  JS2T_CHECK(nparts <= 65535, "colsum: too many rows");
  hipStream_t s = (hipStream_t)stream;
  if (dt == JS2T_BF16 && cols % 8 == 0 && (((uintptr_t)x) & 15) == 0) {
    hipLaunchKernelGGL(colsum_partial_vec_kernel, dim3(cdiv(cols, 512), (unsigned)nparts), dim3(256), 0, s, (const uint16_t*)x,
                       partial, rows, cols);
  } else {
    const dim3 g1(cdiv(cols, 64), (unsigned)nparts);
    DISPATCH_DT(dt, T, hipLaunchKernelGGL((colsum_partial_kernel<T>), g1, dim3(256), 0, s, (const T*)x, partial, rows, cols));
  }
  JS2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(cols, 64)), dim3(256), 0, s, partial, out, nparts, cols, accumulate);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_conv_weight_pack(const float* w, void* wp, int wp_dt, int64_t cout, int64_t cin, int64_t k,
                                     js2t_stream stream) {
  JS2T_CHECK(w && wp && cout > 0 && cin > 0 && k > 0, "conv_weight_pack: bad arguments");
  DISPATCH_DT(wp_dt, D, hipLaunchKernelGGL((conv_pack_kernel<D>), dim3(ew_grid(cout * cin * k)), dim3(EW_THREADS), 0,
                                           (hipStream_t)stream, w, (D*)wp, cout, cin, k));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
extern "C" int js2t_conv_weight_unpack_grad(const float* dwp_t, float* dw, int64_t cout, int64_t cin, int64_t k,
                                            int accumulate, js2t_stream stream) {
  JS2T_CHECK(dwp_t && dw && cout > 0 && cin > 0 && k > 0, "conv_weight_unpack_grad: bad arguments");
  hipLaunchKernelGGL(conv_unpack_grad_kernel, dim3(ew_grid(cout * cin * k)), dim3(EW_THREADS), 0, (hipStream_t)stream, dwp_t,
                     dw, cout, cin, k, accumulate);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
extern "C" int js2t_col2im(const void* dcol, void* dx, int64_t B, int64_t tin, int64_t tout, int64_t C, int64_t K,
                           int64_t stride, int64_t pad, int dt, js2t_stream stream) {
  if (B * tin * C == 0) return JS2T_OK;
  JS2T_CHECK(dcol && dx && stride > 0 && K > 0, "col2im: bad arguments");
  if (dt == JS2T_BF16 && (C & 7) == 0 && ((((uintptr_t)dcol) | ((uintptr_t)dx)) & 15) == 0) {
    hipLaunchKernelGGL(col2im_vec_kernel, dim3(ew_grid(B * tin * (C >> 3))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       (const uint16_t*)dcol, (uint16_t*)dx, B, tin, tout, C, K, stride, pad);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  DISPATCH_DT(dt, T, hipLaunchKernelGGL((col2im_kernel<T>), dim3(ew_grid(B * tin * C)), dim3(EW_THREADS), 0,
                                        (hipStream_t)stream, (const T*)dcol, (T*)dx, B, tin, tout, C, K, stride, pad));
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_im2col(const void* x, void* col, int64_t B, int64_t tin, int64_t tout, int64_t C, int64_t K, int64_t stride,
                           int64_t pad, int dt, js2t_stream stream) {
  if (B * tout * C * K == 0) return JS2T_OK;
  JS2T_CHECK(x && col && stride > 0 && K > 0 && tin > 0, "im2col: bad arguments");
  JS2T_CHECK(dt == JS2T_F32 || dt == JS2T_BF16, "im2col: bad dtype");
  const int vec = dt == JS2T_BF16 ? 8 : 4;
  JS2T_CHECK(C % vec == 0 && ((((uintptr_t)x) | ((uintptr_t)col)) & 15) == 0, "im2col: channels must fill 16-byte pieces, buffers 16-byte aligned");
  const int64_t work = B * tout * K * (C / vec);
  if (dt == JS2T_BF16)
    hipLaunchKernelGGL((im2col_kernel<uint16_t, 8>), dim3(ew_grid(work)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint16_t*)col, B, tin, tout, C, K, stride, pad);
  else
    hipLaunchKernelGGL((im2col_kernel<float, 4>), dim3(ew_grid(work)), dim3(EW_THREADS), 0, (hipStream_t)stream, (const float*)x,
                       (float*)col, B, tin, tout, C, K, stride, pad);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// ------------------------------------------------------------------------------------------------
// Packed rows of a ragged batch (encoders.py:348-373: positions behind an utterance's sub-sampled length are dead): the encoder
// stack runs on the sum(T'_i) live rows only.  pack: dst[seg[b] + t] = src[b*T + t] for t < len_b, rows seg[B] .. rows_out of dst
// zeroed (the caller rounds the packed row count up to a launch-friendly size); unpack: dst[b*T + t] = t < len_b ?
// src[seg[b] + t] : 0.  16-byte pieces, eight rows per block; blockIdx.y = batch entry (pack: entry B = the zeroed tail).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int PK_ROWS = 8;
template <bool PACK>
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int32_t* __restrict__ seg,
                                                        int B, int T, int64_t rows_out, int chunks) {
  const int b = blockIdx.y, t0 = blockIdx.x * PK_ROWS;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
  if (PACK && b == B) {  // the tail behind the last entry, however long: the slab's blocks stride over it
    for (int64_t r0 = (int64_t)seg[B] + t0; r0 < rows_out; r0 += (int64_t)gridDim.x * PK_ROWS)
      for (int i = threadIdx.x; i < PK_ROWS * chunks; i += 256) {
        const int64_t r = r0 + i / chunks;
        if (r < rows_out) dst[r * chunks + i % chunks] = zero;
      }
    return;
  }
  const int s0 = seg[b], len = seg[b + 1] - s0;
  if (PACK && t0 >= len) return;
  for (int i = threadIdx.x; i < PK_ROWS * chunks; i += 256) {
    const int t = t0 + i / chunks, c = i % chunks;
    if (t >= T) break;
    if (PACK) {
      if (t < len) dst[((int64_t)s0 + t) * chunks + c] = src[((int64_t)b * T + t) * chunks + c];
    } else {
      dst[((int64_t)b * T + t) * chunks + c] = t < len ? src[((int64_t)s0 + t) * chunks + c] : zero;
    }
  }
}
}  // namespace

extern "C" int js2t_pack_rows(const void* src, void* dst, const int32_t* seg, int32_t B, int32_t T_, int64_t rows_out, int64_t row_bytes,
                              int pack, js2t_stream stream) {
  JS2T_CHECK(src && dst && seg && B > 0 && T_ > 0 && row_bytes > 0 && (row_bytes % 16) == 0, "pack_rows: bad arguments");
  JS2T_CHECK(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "pack_rows: buffers must be 16-byte aligned");
  JS2T_CHECK(B < 65535 && (!pack || rows_out > 0), "pack_rows: bad sizes");
  const int chunks = (int)(row_bytes / 16);
  hipStream_t s = (hipStream_t)stream;
  if (pack == 2) {  // only the tail: rows seg[B] .. rows_out of dst zeroed (src unused)
    hipLaunchKernelGGL(pack_rows_kernel<true>, dim3(cdiv(T_, PK_ROWS), 1), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, seg + B, 0, T_,
                       rows_out, chunks);
  } else if (pack) {
    hipLaunchKernelGGL(pack_rows_kernel<true>, dim3(cdiv(T_, PK_ROWS), B + 1), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, seg, B, T_,
                       rows_out, chunks);
  } else {
    hipLaunchKernelGGL(pack_rows_kernel<false>, dim3(cdiv(T_, PK_ROWS), B), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, seg, B, T_,
                       rows_out, chunks);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_subsample_lengths_mask(const int64_t* lengths, int64_t* out_lengths, uint8_t* mask, int64_t B,
                                           int64_t T_out, const int32_t* kernel_sizes, int32_t n_layers, js2t_stream stream) {
  JS2T_CHECK(lengths && out_lengths && mask && B > 0 && T_out > 0, "subsample_lengths_mask: bad arguments");
  JS2T_CHECK(n_layers >= 0 && n_layers <= 4 && (n_layers == 0 || kernel_sizes), "subsample_lengths_mask: 0..4 conv layers");
  int ks[4] = {1, 1, 1, 1};
  for (int i = 0; i < n_layers; ++i) ks[i] = kernel_sizes[i];
  hipLaunchKernelGGL(subsample_len_mask_kernel, dim3(ew_grid(B * T_out)), dim3(EW_THREADS), 0, (hipStream_t)stream, lengths,
                     out_lengths, mask, B, T_out, ks[0], ks[1], ks[2], ks[3], n_layers);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_transpose_groups(const void* src, void* dst, const int64_t* table, int32_t n_groups, int64_t total_tiles,
                                     js2t_stream stream) {
  if (n_groups == 0 || total_tiles == 0) return JS2T_OK;
  JS2T_CHECK(src && dst && table && n_groups > 0 && total_tiles > 0 && total_tiles < 0x7fffffff, "transpose_groups: bad arguments");
  hipLaunchKernelGGL(transpose_groups_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src,
                     (uint16_t*)dst, table, n_groups);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// ---------------------------------------------------------------- e4m3 quantisation (fp8 forward mode, BASELINE config 5)
namespace {
// max |x| over a tensor: block maxima land in *out through an unsigned atomic max (non-negative floats order like their bits)
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  constexpr int V = 16 / sizeof(T);
  for (int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * V; i < n; i += (int64_t)gridDim.x * 256 * V) {
    if (i + V <= n) {
      const uint4 r = *(const uint4*)(x + i);
      if constexpr (sizeof(T) == 2) {
        const uint32_t wds[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaxf(fabsf(__uint_as_float(wds[k] << 16)), fabsf(__uint_as_float(wds[k] & 0xffff0000u))));
      } else {
        m = fmaxf(m, fmaxf(fmaxf(fabsf(__uint_as_float(r.x)), fabsf(__uint_as_float(r.y))),
                           fmaxf(fabsf(__uint_as_float(r.z)), fabsf(__uint_as_float(r.w)))));
      }
    } else {
      for (int64_t j = i; j < n; ++j) m = fmaxf(m, fabsf(io<T>::ld(x + j)));
    }
  }
  m = block_max(m, red);
  // same-address atomics retire one per ~12 ns chip-wide: post only maxima that beat what is already there (a stale read is
  // fine, the atomic decides).  NaN inputs would poison the scale: not handled
  if (threadIdx.x == 0 && __float_as_uint(m) > *(volatile unsigned int*)out) atomicMax((unsigned int*)out, __float_as_uint(m));
}

// 8 consecutive elements as floats: one 16-byte (bf16) or two 16-byte (f32) loads when the whole group is in range and aligned
template <typename T>
__device__ __forceinline__ void load8f(const T* __restrict__ x, int64_t i, int64_t n, float (&v)[8]) {
  if (i + 8 <= n && (((uintptr_t)(x + i)) & 15) == 0) {
    if constexpr (sizeof(T) == 2) {
      const uint4 r = *(const uint4*)(x + i);
      const uint32_t wds[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[2 * k] = __uint_as_float(wds[k] << 16);
        v[2 * k + 1] = __uint_as_float(wds[k] & 0xffff0000u);
      }
    } else {
      const float4 a = *(const float4*)(x + i), b = *(const float4*)(x + i + 4);
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = i + k < n ? io<T>::ld(x + i + k) : 0.f;
  }
}

// y = e4m3(x * 448 / amax) (round to nearest even, clamped to +-448); scale_out = amax / 448 * (*mul or 1)
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, uint8_t* __restrict__ y, int64_t n,
                                                          const float* __restrict__ amax, const float* __restrict__ mul,
                                                          float* __restrict__ scale_out) {
  const float am = *amax;
  const float inv = am > 0.f ? 448.f / am : 0.f;
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = (am > 0.f ? am / 448.f : 1.f) * (mul ? *mul : 1.f);
  for (int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
    float v[8];
    load8f(x, i, n, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fminf(fmaxf(v[k] * inv, -448.f), 448.f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    if (i + 8 <= n) {
      *(uint2*)(y + i) = make_uint2((uint32_t)lo, (uint32_t)hi);
    } else {
      const uint32_t wds[2] = {(uint32_t)lo, (uint32_t)hi};
      for (int k = 0; i + k < n; ++k) y[i + k] = (uint8_t)(wds[k >> 2] >> (8 * (k & 3)));
    }
  }
}
}  // namespace

extern "C" int js2t_absmax(const void* x, int dt, int64_t n, float* out, js2t_stream stream) {
  JS2T_CHECK(x && out && n > 0 && (dt == JS2T_F32 || dt == JS2T_BF16), "absmax: bad arguments");
  JS2T_CHECK((((uintptr_t)x) & 15) == 0, "absmax: input must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
  if (e != hipSuccess) {
    js2t_set_error("absmax: %s", hipGetErrorString(e));
    return JS2T_ERR_LAUNCH;
  }
  const int64_t per = dt == JS2T_BF16 ? 8 : 4;
  int64_t grid = (n / per + 255) / 256;
  grid = grid < 1 ? 1 : (grid > 512 ? 512 : grid);
  if (dt == JS2T_BF16)
    hipLaunchKernelGGL(absmax_kernel<uint16_t>, dim3((unsigned)grid), dim3(256), 0, s, (const uint16_t*)x, n, out);
  else
    hipLaunchKernelGGL(absmax_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, (const float*)x, n, out);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_quantize_fp8(const void* x, int dt, void* y, int64_t n, const float* amax, const float* mul, float* scale_out,
                                 js2t_stream stream) {
  JS2T_CHECK(x && y && amax && n > 0 && (dt == JS2T_F32 || dt == JS2T_BF16), "quantize_fp8: bad arguments");
  JS2T_CHECK((((uintptr_t)y) & 7) == 0, "quantize_fp8: output must be 8-byte aligned");
  int64_t grid = (n / 8 + 255) / 256;
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  if (dt == JS2T_BF16)
    hipLaunchKernelGGL(quantize_fp8_kernel<uint16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint8_t*)y, n, amax, mul, scale_out);
  else
    hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (uint8_t*)y, n,
                       amax, mul, scale_out);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// ---------------------------------------------------------------- e4m3 quantisation with DELAYED scaling: one pass
// state[0] = scale S in use (amax / 448 of an earlier call), state[1] = running max |x| of this call (uint bits),
// state[2] = arrival ticket.  y = e4m3(clamp(x / S)); the block that finishes last turns the collected maximum into the
// next call's scale and clears the two counters - every other block has read S by then.  Replays of a captured hipGraph
// therefore keep adapting the scale with no host involvement.  Values beyond the stale maximum saturate at +-448.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_delayed_kernel(const T* __restrict__ x, uint8_t* __restrict__ y, int64_t n,
                                                                  float* __restrict__ state, const float* __restrict__ mul,
                                                                  float* __restrict__ scale_out) {
  __shared__ float red[4];
  const float S = state[0];
  const float inv = S > 0.f ? 1.f / S : 0.f;
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = (S > 0.f ? S : 1.f) * (mul ? *mul : 1.f);
  float m = 0.f;
  for (int64_t i = (blockIdx.x * (int64_t)256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
    float v[8];
    load8f(x, i, n, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      m = fmaxf(m, fabsf(v[k]));
      v[k] = fminf(fmaxf(v[k] * inv, -448.f), 448.f);
    }
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    if (i + 8 <= n) {
      *(uint2*)(y + i) = make_uint2((uint32_t)lo, (uint32_t)hi);
    } else {
      const uint32_t wds[2] = {(uint32_t)lo, (uint32_t)hi};
      for (int k = 0; i + k < n; ++k) y[i + k] = (uint8_t)(wds[k >> 2] >> (8 * (k & 3)));
    }
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) {
    unsigned int* st = (unsigned int*)state;
    // post the maximum (only if it beats what is there: see absmax_kernel) with a RETURNING atomic whose result is consumed: it
    // has been performed at the L2 before the ticket below is drawn, so no fence (buffer_wbl2 + invalidate, ~3.5 us per
    // block) is needed - both words are only ever touched by device-scope atomics
    unsigned int seen = 0u;
    if (__float_as_uint(m) > *(volatile unsigned int*)(st + 1)) seen = atomicMax(st + 1, __float_as_uint(m));
    asm volatile("" ::"v"(seen));
    if (atomicAdd(st + 2, 1u) == gridDim.x - 1) {  // last block: every other one has read S and posted its maximum
      const float am = __uint_as_float(atomicExch(st + 1, 0u));
      if (am > 0.f) state[0] = am / 448.f;
      st[2] = 0u;
    }
  }
}
}  // namespace

extern "C" int js2t_quantize_fp8_delayed(const void* x, int dt, void* y, int64_t n, float* state, const float* mul, float* scale_out,
                                         js2t_stream stream) {
  JS2T_CHECK(x && y && state && n > 0 && (dt == JS2T_F32 || dt == JS2T_BF16), "quantize_fp8_delayed: bad arguments");
  JS2T_CHECK((((uintptr_t)y) & 7) == 0 && (((uintptr_t)state) & 15) == 0, "quantize_fp8_delayed: misaligned output / state");
  int64_t grid = (n / 8 + 255) / 256;
  grid = grid < 1 ? 1 : (grid > 512 ? 512 : grid);  // one arrival-ticket atomic per block: keep the blocks few
  if (dt == JS2T_BF16)
    hipLaunchKernelGGL(quantize_fp8_delayed_kernel<uint16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (uint8_t*)y, n, state, mul, scale_out);
  else
    hipLaunchKernelGGL(quantize_fp8_delayed_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (uint8_t*)y, n, state, mul, scale_out);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
