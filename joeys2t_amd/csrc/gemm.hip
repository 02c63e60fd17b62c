// MFMA GEMM with fused epilogue for gfx950.  Two kernels behind one C entry point (js2t_gemm):
//
//  * gemm_bf16_kernel<TA,TB> — the hot kernel.  128x128x64 block tile, 4 waves (2x2), each wave a 64x64
//    sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  Operands are staged global -> registers ->
//    LDS (issue-early / write-late, double-buffered LDS, one barrier per K tile).  A k-contiguous operand
//    tile is kept as [row][64 k] with a 16-byte-chunk XOR swizzle (conflict-free ds_read_b128); an operand
//    whose reduction index is the slow one in memory (trans_a / trans_b) is kept as [k][row] and read
//    with ds_read_b64_tr_b16, so NN / TN products (dgrad, wgrad, P.V) need no transposed copies.
//    The MFMA is issued "swapped" (A-operand = B tile, B-operand = A tile) so that every lane ends up with
//    4 consecutive output columns of one row -> 8/16-byte epilogue accesses.
//
//  * gemm_generic_kernel — any shape / alignment / dtype, v_mfma_f32_16x16x4_f32 (exact f32 FMA chain).
//    This is the fp32 "parity mode" kernel and the fallback for bf16 operands the fast kernel rejects.
//
// Reference call sites replaced: see include/joeys2t_hip.h (js2t_gemm).
#include "gemm_shared.hpp"
#include <type_traits>

namespace {

bool g_force_regstage = false;  // test hook: route non-conv bf16 GEMMs to the register-staged kernel
bool g_force_w256 = false;      // test hook: take the 256x256 kernel for every product it can run, whatever the size

// ------------------------------------------------------------------------------------------------
// operand addressing shared by both kernels
// ------------------------------------------------------------------------------------------------
// Element offset of A at (slow, fast): slow runs along lda, fast is contiguous.  In conv mode slow is
// the (b, t_out) output-frame index and fast is kw*C + c; returns false for zero padding.
__device__ __forceinline__ bool a_elem_offset(const js2t_gemm_desc& d, int slow, int fast, int64_t& off) {
  if (!d.conv) {
    off = (int64_t)slow * d.lda + fast;
    return true;
  }
  const int b = slow / d.conv_tout, t = slow - b * d.conv_tout;
  const int kw = fast / d.conv_c;
  const int tin = t * d.conv_stride - d.conv_pad + kw;
  off = ((int64_t)b * d.conv_tin + (t * d.conv_stride - d.conv_pad)) * d.conv_c + fast;
  return tin >= 0 && tin < d.conv_tin;
}

// Epilogue for 4 consecutive columns n..n+3 of row m (see header for the order of operations).
__device__ __forceinline__ void gemm_epilogue4(const js2t_gemm_desc& d, int z, int64_t c_boff, int m, int n,
                                               f32x4_t acc, float alpha) {
  if (m >= d.M || n >= d.N) return;
  const int nv = min(4, d.N - n);
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = acc[i] * alpha;

  if (d.bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) v[i] += d.bias[n + i];
  }
  const int64_t coff = c_boff + (int64_t)m * d.ldc + n;
  if (d.preact) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) st_elem(d.preact, coff + i, d.dtype_c, v[i]);
  }
  if (d.act != JS2T_ACT_NONE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_apply(v[i], d.act);
  }
  if (d.dropout_p > 0.f) {
    const uint32_t keep = dropout_keep4(d.rng_state, d.rng_stream, (uint32_t)(z * d.M + m), (uint32_t)(n >> 2),
                                        d.dropout_p);
    const float sc = 1.f / (1.f - d.dropout_p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((keep >> i) & 1u) ? v[i] * sc : 0.f;
  }
  if (d.residual) {
    const int64_t roff = (int64_t)m * d.ldr + n;  // residual / gate are only accepted for batch == 1
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) v[i] += d.res_scale * ld_elem(d.residual, roff + i, d.dtype_c);
  }
  if (d.gate) {
    const int64_t goff = (int64_t)m * d.ldg + n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) v[i] = ld_elem(d.gate, goff + i, d.dtype_c) > 0.f ? v[i] * d.gate_scale : 0.f;
  }
  if (d.beta != 0.f) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) v[i] += d.beta * ld_elem(d.C, coff + i, d.dtype_c);
  }
  if (nv == 4 && ((coff & 3) == 0)) {
    if (d.dtype_c == JS2T_F32) {
      *(float4*)((float*)d.C + coff) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint2 pk;
      pk.x = pack_bf16x2(v[0], v[1]);
      pk.y = pack_bf16x2(v[2], v[3]);
      *(uint2*)((uint16_t*)d.C + coff) = pk;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < nv) st_elem(d.C, coff + i, d.dtype_c, v[i]);
  }
}

__device__ __forceinline__ void batch_offsets(const js2t_gemm_desc& d, int z, int64_t& ao, int64_t& bo, int64_t& co) {
  const int zo = z / d.batch_inner, zi = z - zo * d.batch_inner;
  ao = (int64_t)zo * d.a_stride_o + (int64_t)zi * d.a_stride_i;
  bo = (int64_t)zo * d.b_stride_o + (int64_t)zi * d.b_stride_i;
  co = (int64_t)zo * d.c_stride_o + (int64_t)zi * d.c_stride_i;
}


// ------------------------------------------------------------------------------------------------
// generic kernel: 64x64x16 tile, f32 MFMA 16x16x4, element-wise bounds-checked staging
// ------------------------------------------------------------------------------------------------
constexpr int G_BM = 64, G_BN = 64, G_BK = 16, G_LD = 80;

__global__ __launch_bounds__(256) void gemm_generic_kernel(js2t_gemm_desc d, int tiles_m, int tiles_n) {
  __shared__ float As[G_BK][G_LD];
  __shared__ float Bs[G_BK][G_LD];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int z = blockIdx.y;
  const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * G_BM, n0 = (lid % tiles_n) * G_BN;
  int64_t ao, bo, co;
  batch_offsets(d, z, ao, bo, co);
  const int wm = w >> 1, wn = w & 1;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < d.K; k0 += G_BK) {
    // stage A
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int m, k;
      if (!d.trans_a) { k = t & 15; m = (t >> 4) + 16 * r; }
      else            { m = t & 63; k = (t >> 6) + 4 * r; }
      float v = 0.f;
      const int gm = m0 + m, gk = k0 + k;
      if (gm < d.M && gk < d.K) {
        int64_t off;
        const bool ok = d.trans_a ? a_elem_offset(d, gk, gm, off) : a_elem_offset(d, gm, gk, off);
        if (ok) v = ld_elem(d.A, ao + off, d.dtype_ab);
      }
      As[k][m] = v;
    }
    // stage B
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n, k;
      if (!d.trans_b) { k = t & 15; n = (t >> 4) + 16 * r; }
      else            { n = t & 63; k = (t >> 6) + 4 * r; }
      float v = 0.f;
      const int gn = n0 + n, gk = k0 + k;
      if (gn < d.N && gk < d.K) {
        const int64_t off = d.trans_b ? (int64_t)gk * d.ldb + gn : (int64_t)gn * d.ldb + gk;
        v = ld_elem(d.B, bo + off, d.dtype_ab);
      }
      Bs[k][n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < G_BK / 4; ++kk) {
      const int kr = kk * 4 + (lane >> 4);
      float bn[2], am[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bn[j] = Bs[kr][wn * 32 + 16 * j + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i) am[i] = As[kr][wm * 32 + 16 * i + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bn[j], am[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 32 + 16 * i + (lane & 15);
      const int n = n0 + wn * 32 + 16 * j + 4 * (lane >> 4);
      gemm_epilogue4(d, z, co, m, n, acc[i][j], alpha);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 fast kernel
// ------------------------------------------------------------------------------------------------
constexpr int F_BM = 128, F_BN = 128, F_BK = 64;
constexpr int KC_ROW_BYTES = 128;          // [row][64 k] bf16
constexpr int TR_ROW_BYTES = 288;          // [k][128 rows] bf16 + 32 B pad
constexpr int KC_TILE_BYTES = 128 * KC_ROW_BYTES;  // 16384
constexpr int TR_TILE_BYTES = 64 * TR_ROW_BYTES;   // 18432

// 8 bf16 of operand X at (slow, fast..fast+7); zero outside [0,slow_max) x [0,fast_max).
template <bool IS_A>
__device__ __forceinline__ uint4 load8(const js2t_gemm_desc& d, const uint16_t* base, int64_t ld, int slow,
                                       int fast, int slow_max, int fast_max) {
  uint4 r = make_uint4(0u, 0u, 0u, 0u);
  if (slow >= slow_max || fast >= fast_max) return r;
  int64_t off;
  bool ok = true;
  if (IS_A) ok = a_elem_offset(d, slow, fast, off);
  else off = (int64_t)slow * ld + fast;
  if (!ok) return r;
  if (fast + 8 <= fast_max) return *(const uint4*)(base + off);
  uint16_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = (fast + i < fast_max) ? base[off + i] : (uint16_t)0;
  r.x = e[0] | ((uint32_t)e[1] << 16); r.y = e[2] | ((uint32_t)e[3] << 16);
  r.z = e[4] | ((uint32_t)e[5] << 16); r.w = e[6] | ((uint32_t)e[7] << 16);
  return r;
}

// global -> registers for one 128 x 64 operand tile (4 x 16 B per thread)
template <bool TR, bool IS_A>
__device__ __forceinline__ void stage_load(const js2t_gemm_desc& d, const uint16_t* base, int64_t ld, int rowbase,
                                           int k0, int rows_max, int K, int t, uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!TR) {
      const int row = (t >> 3) + 32 * i, c = t & 7;
      r[i] = load8<IS_A>(d, base, ld, rowbase + row, k0 + 8 * c, rows_max, K);
    } else {
      const int kr = (t >> 4) + 16 * i, c = t & 15;
      r[i] = load8<IS_A>(d, base, ld, k0 + kr, rowbase + 8 * c, K, rows_max);
    }
  }
}
// registers -> LDS
template <bool TR>
__device__ __forceinline__ void stage_store(unsigned char* tile, int t, const uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!TR) {
      const int row = (t >> 3) + 32 * i, c = t & 7;
      *(uint4*)(tile + row * KC_ROW_BYTES + ((c ^ (row & 7)) << 4)) = r[i];
    } else {
      const int kr = (t >> 4) + 16 * i, c = t & 15;
      *(uint4*)(tile + kr * TR_ROW_BYTES + ((c << 4) ^ (((kr >> 3) & 1) << 7))) = r[i];
    }
  }
}
// LDS -> MFMA fragment: 8 k-values (kk*32 + 8*(lane>>4) + 0..7) of tile row (sub + (lane&15))
template <bool TR>
__device__ __forceinline__ bf16x8_t frag_load(const unsigned char* tile, int sub, int kk, int lane) {
  if (!TR) {
    const int row = sub + (lane & 15), c = kk * 4 + (lane >> 4);
    return *(const bf16x8_t*)(tile + row * KC_ROW_BYTES + ((c ^ (row & 7)) << 4));
  } else {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3;
    const int kb = kk * 32 + 8 * (lane >> 4);
    const int colb = (sub + 4 * p) * 2;
    const int k1 = kb + q, k2 = kb + 4 + q;
    typedef __attribute__((address_space(3))) s16x4_t* lds_p;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_p)(tile + k1 * TR_ROW_BYTES + (colb ^ (((k1 >> 3) & 1) << 7))));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_p)(tile + k2 * TR_ROW_BYTES + (colb ^ (((k2 >> 3) & 1) << 7))));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// SPLITK instantiations reduce one K slice per blockIdx.z and add their tile into a zero-filled f32 C with atomics;
// that path is kept out of the plain instantiations (and off the descriptor: the pointer arrives as its own argument).
template <bool TA, bool TB, bool SPLITK>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(js2t_gemm_desc d, int tiles_m, int tiles_n, float* c_atomic,
                                                           int split_k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_BYTES = TA ? TR_TILE_BYTES : KC_TILE_BYTES;
  constexpr int B_BYTES = TB ? TR_TILE_BYTES : KC_TILE_BYTES;
  constexpr int STAGE = A_BYTES + B_BYTES;

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int z = blockIdx.y;
  const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * F_BM, n0 = (lid % tiles_n) * F_BN;
  int64_t ao, bo, co;
  batch_offsets(d, z, ao, bo, co);
  const uint16_t* Ab = (const uint16_t*)d.A + ao;
  const uint16_t* Bb = (const uint16_t*)d.B + bo;
  const int wm = w >> 1, wn = w & 1;
  // conv mode: A's "rows" are output frames; M (or K for trans_a) bounds still apply through M/K.
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  const int nk_all = (d.K + F_BK - 1) / F_BK;
  const int per = SPLITK ? (nk_all + split_k - 1) / split_k : nk_all;
  const int kt0 = SPLITK ? blockIdx.z * per : 0;
  const int nk = min(per, nk_all - kt0);
  if (SPLITK && nk <= 0) return;
  stage_load<TA, true>(d, Ab, d.lda, m0, kt0 * F_BK, d.M, d.K, t, ra);
  stage_load<TB, false>(d, Bb, d.ldb, n0, kt0 * F_BK, d.N, d.K, t, rb);
  stage_store<TA>(smem, t, ra);
  stage_store<TB>(smem + A_BYTES, t, rb);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      stage_load<TA, true>(d, Ab, d.lda, m0, (kt0 + kt + 1) * F_BK, d.M, d.K, t, ra);
      stage_load<TB, false>(d, Bb, d.ldb, n0, (kt0 + kt + 1) * F_BK, d.N, d.K, t, rb);
    }
    const unsigned char* At = smem + cur * STAGE;
    const unsigned char* Bt = At + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t fn[4], fm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fn[j] = frag_load<TB>(Bt, wn * 64 + 16 * j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) fm[i] = frag_load<TA>(At, wm * 64 + 16 * i, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[j], fm[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      unsigned char* An = smem + (cur ^ 1) * STAGE;
      stage_store<TA>(An, t, ra);
      stage_store<TB>(An + A_BYTES, t, rb);
    }
    __syncthreads();
    cur ^= 1;
  }
  if constexpr (SPLITK) {
    const int M = d.M, N = d.N;
    const int64_t ldc = d.ldc;
    const float alpha = d.alpha;
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat* cg = (gfloat*)(c_atomic + co);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + 16 * i + (lane & 15);
        const int n = n0 + wn * 64 + 16 * j + 4 * (lane >> 4);
        if (m < M) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) __builtin_amdgcn_global_atomic_fadd_f32(cg + (int64_t)m * ldc + n + r, acc[i][j][r] * alpha);
        }
      }
  } else {
    const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + 16 * i + (lane & 15);
        const int n = n0 + wn * 64 + 16 * j + 4 * (lane >> 4);
        gemm_epilogue4(d, z, co, m, n, acc[i][j], alpha);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 fast kernel, second generation: LDS-DMA staging (global_load_lds_dwordx4) + LDS-staged epilogue
// ------------------------------------------------------------------------------------------------
// Same 128x128x64 tiling / wave layout as above, but
//  * operand tiles go HBM -> LDS directly (no VGPR round trip, no ds_write): every wave-instruction lands 1 KiB
//    contiguously, so the bank-conflict swizzles are applied to the per-lane SOURCE address and undone by the reads;
//    tile t+1 streams in while the MFMAs of tile t run (2 LDS stages, one s_waitcnt vmcnt(0) + barrier per K tile);
//  * a transposed-operand tile is a pad-free [64 k][128 rows] image whose 32-byte column groups are XORed with
//    g(k) = (k&3) | ((k>>3)&1)<<2, which makes the ds_read_b64_tr_b16 fragment reads conflict-free;
//  * the accumulators are staged through LDS so the epilogue reads/writes whole 16-byte row segments
//    (bias / residual / gate / dropout / C all coalesced) instead of 8-byte scattered pieces.
// Rows beyond M/N are clamped to the last valid row (their products are never stored); a partial last K tile is
// zero-filled in LDS after the DMA has landed.  Implicit-conv operands stay on the register-staged kernel above.
typedef __attribute__((address_space(1))) const void g_cvoid;
typedef __attribute__((address_space(3))) void l_void;

__device__ __forceinline__ int tr_g(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

template <bool TR>
__device__ __forceinline__ bf16x8_t frag_load2(const unsigned char* tile, int sub, int kk, int lane) {
  if (!TR) {
    const int row = sub + (lane & 15), c = kk * 4 + (lane >> 4);
    return *(const bf16x8_t*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
  } else {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3;
    const int kb = kk * 32 + 8 * (lane >> 4);
    const int colb = (sub + 4 * p) * 2;
    const int k1 = kb + q, k2 = kb + 4 + q;
    typedef __attribute__((address_space(3))) s16x4_t* lds_p;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + k1 * 256 + (colb ^ (tr_g(k1) << 5))));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tile + k2 * 256 + (colb ^ (tr_g(k2) << 5))));
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// per-lane source element offsets of the NP DMA pieces this wave issues for one operand tile of ROWS rows
// (kc image: ROWS x 64 k, NP = ROWS/32; tr image: 64 k x 128 rows, NP = 4)
// XOR key of a [row][64 k] image row.  Natural fragment rows (16 consecutive rows per read) are conflict-free with
// row & 7; the column-permuted B fragments (rows 4j + (l&3) of four 16-row blocks, see dma_gemm_block) put blocks 0/3 and
// 1/2 in one ds_read_b128 lane group, which row & 7 maps to the same 16-byte slots (2-way): their key takes bit 2 from
// the block index instead.
template <bool PERMB>
__device__ __forceinline__ int kc_key(int row) {
  return PERMB ? ((row & 3) | (((row >> 4) & 1) << 2)) : (row & 7);
}

template <bool TR, int NP, bool PERMB = false>
__device__ __forceinline__ void dma_offsets(int64_t ld, int rowbase, int rows_max, int k0, int K, int t, int64_t (&off)[NP]) {
  const int w = t >> 6, lane = t & 63;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int p = (w * NP + q) * 64 + lane;  // 16-byte slot index inside the tile
    if (!TR) {
      const int row = p >> 3, slot = p & 7;
      int chunk = slot ^ kc_key<PERMB>(row);
      if (k0 + chunk * 8 >= K) chunk = 0;  // clamped; zero-filled afterwards
      off[q] = (int64_t)min(rowbase + row, rows_max - 1) * ld + k0 + chunk * 8;
    } else {
      const int kr = p >> 4, slot = p & 15;
      int c = rowbase + ((slot ^ (tr_g(kr) << 1)) << 3);
      if (c >= rows_max) c = rowbase;  // rows that are never stored
      off[q] = (int64_t)min(k0 + kr, K - 1) * ld + c;
    }
  }
}
// HIDDEN: issue through lds_dma16 (kernels whose fragments come from transposing reads, see there)
template <int NP, bool HIDDEN = false>
__device__ __forceinline__ void dma_issue(const uint16_t* base, const int64_t (&off)[NP], unsigned char* tile, int t) {
  const int w = HIDDEN ? __builtin_amdgcn_readfirstlane(t >> 6) : t >> 6;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    if constexpr (HIDDEN)
      lds_dma16(base + off[q], tile + (w * NP + q) * 1024);
    else
      __builtin_amdgcn_global_load_lds((g_cvoid*)(base + off[q]), (l_void*)(tile + (w * NP + q) * 1024), 16, 0, 0);
  }
}
// zero the LDS slots of a partial last K tile (k >= k_lim)
template <bool TR, int NP, bool PERMB = false>
__device__ __forceinline__ void dma_zero_tail(unsigned char* tile, int k_lim, int t) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int p = ((t >> 6) * NP + q) * 64 + (t & 63);  // the slots this wave's own DMA instructions filled
    if (!TR) {
      const int row = p >> 3, slot = p & 7;
      if (((slot ^ kc_key<PERMB>(row)) << 3) >= k_lim) *(uint4*)(tile + p * 16) = z;
    } else {
      if ((p >> 4) >= k_lim) *(uint4*)(tile + p * 16) = z;
    }
  }
}

// epilogue for 8 consecutive columns of one row, values already in registers (see header for the op order).
// Everything that does not depend on the row (bias, dropout key, activation kind) is resolved by the caller once per
// thread; residual / gate rows arrive pre-loaded (packed bf16) when the fast path applies.
__device__ __forceinline__ void unpack_bf16x8(const uint4& r, float (&o)[8]) {
  const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float(rr[i] << 16);
    o[2 * i + 1] = __uint_as_float(rr[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void gemm_epilogue8(const js2t_gemm_desc& d, int z, int64_t c_boff, int m, int n, float (&v)[8],
                                               float alpha, uint32_t drop_key, const float (&bias_r)[8], bool vec_rg,
                                               const uint4& res_pk, const uint4& gate_pk) {
  const int nv = min(8, d.N - n);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + bias_r[i];
  const int64_t coff = c_boff + (int64_t)m * d.ldc + n;
  const bool full = nv == 8;
  if (d.preact) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) st_elem(d.preact, coff + i, d.dtype_c, v[i]);
  }
  if (d.act == JS2T_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  } else if (d.act != JS2T_ACT_NONE) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = act_apply(v[i], d.act);
  }
  if (d.dropout_p > 0.f) {
    const float sc = 1.f / (1.f - d.dropout_p);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t keep = dropout_keep4_key(drop_key, (uint32_t)(z * d.M + m), (uint32_t)((n >> 2) + h), d.dropout_p);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[4 * h + i] = ((keep >> i) & 1u) ? v[4 * h + i] * sc : 0.f;
    }
  }
  if (d.residual) {
    if (vec_rg) {
      float r[8];
      unpack_bf16x8(res_pk, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] += d.res_scale * r[i];
    } else {
      const int64_t roff = (int64_t)m * d.ldr + n;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < nv) v[i] += d.res_scale * ld_elem(d.residual, roff + i, d.dtype_c);
    }
  }
  if (d.gate) {
    if (vec_rg) {
      float r[8];
      unpack_bf16x8(gate_pk, r);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = r[i] > 0.f ? v[i] * d.gate_scale : 0.f;
    } else {
      const int64_t goff = (int64_t)m * d.ldg + n;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < nv) v[i] = ld_elem(d.gate, goff + i, d.dtype_c) > 0.f ? v[i] * d.gate_scale : 0.f;
    }
  }
  if (d.beta != 0.f) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) v[i] += d.beta * ld_elem(d.C, coff + i, d.dtype_c);
  }
  if (full && d.dtype_c == JS2T_BF16 && ((coff & 7) == 0)) {
    uint4 pk;
    pk.x = pack_bf16x2(v[0], v[1]);
    pk.y = pack_bf16x2(v[2], v[3]);
    pk.z = pack_bf16x2(v[4], v[5]);
    pk.w = pack_bf16x2(v[6], v[7]);
    *(uint4*)((uint16_t*)d.C + coff) = pk;
  } else if (full && d.dtype_c == JS2T_F32 && ((coff & 3) == 0)) {
    *(float4*)((float*)d.C + coff) = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)((float*)d.C + coff + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nv) st_elem(d.C, coff + i, d.dtype_c, v[i]);
  }
}

// Tile epilogue shared by the LDS-DMA kernels: accumulators -> LDS -> row-major walk (fused epilogue / split-K atomics).
// The caller guarantees that no DMA into `smem` is outstanding and that every wave is done reading operand tiles
// (the function starts with a barrier).
// PERM: accumulator j, register r of lane group g = lane>>4 is tile column 16g + 4j + r (see the kernel) instead of
// 16j + 4g + r.
template <int BM, bool SPLITK, bool PERM>
__device__ __forceinline__ void dma_tile_epilogue(const js2t_gemm_desc& d, f32x4_t (&acc)[BM / 32][4], unsigned char* smem, int z,
                                                  int64_t co, int m0, int n0, float* c_atomic) {
  constexpr int MI = BM / 32;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int M = d.M, N = d.N;
  if constexpr (SPLITK) {
    // K-slice partial tile -> LDS -> f32 atomics issued as whole 256-byte row segments (one row half per wave
    // instruction): scattered 4-byte atomics run an order of magnitude below the ~1.3 TB/s contiguous atomic rate
    __syncthreads();
    float* Cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ml = wm * (BM / 2) + 16 * i + (lane & 15);
        const int n4 = PERM ? wn * 16 + 4 * (lane >> 4) + j : wn * 16 + 4 * j + (lane >> 4);
        *(f32x4_t*)(Cs + ml * 128 + ((n4 ^ (ml & 7)) << 2)) = acc[i][j];
      }
    __syncthreads();
    const int64_t ldc = d.ldc;
    const float alpha = d.alpha;
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat* cg = (gfloat*)(c_atomic + co);
    for (int idx = w; idx < 2 * BM; idx += 4) {
      const int ml = idx >> 1, nl = (idx & 1) * 64 + lane;
      const int m = m0 + ml, n = n0 + nl;
      if (m < M && n < N) {
        const float v = Cs[ml * 128 + ((((nl >> 2) ^ (ml & 7)) << 2) | (nl & 3))];
        __builtin_amdgcn_global_atomic_fadd_f32(cg + (int64_t)m * ldc + n, v * alpha);
      }
    }
  } else {
    // stage the 128x128 f32 tile in LDS ([row][32 float4 slots], slot ^= row&7), then walk it row-major
    __syncthreads();
    float* Cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ml = wm * (BM / 2) + 16 * i + (lane & 15);
        const int n4 = PERM ? wn * 16 + 4 * (lane >> 4) + j : wn * 16 + 4 * j + (lane >> 4);
        *(f32x4_t*)(Cs + ml * 128 + ((n4 ^ (ml & 7)) << 2)) = acc[i][j];
      }
    __syncthreads();
    const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f);
    const uint32_t drop_key = d.dropout_p > 0.f ? dropout_key(d.rng_state, d.rng_stream) : 0u;
    // this thread owns the same 8 columns in all passes (16 rows per pass): the bias is fetched once
    const int c8 = t & 15, r16 = t >> 4, n = n0 + c8 * 8;
    float bias_r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bias_r[i] = (d.bias && n + i < N) ? d.bias[n + i] : 0.f;
    constexpr int NPASS = BM / 16;
    // Fast path (block-uniform): whole tile inside N, 16-byte row segments, and only the terms the train step
    // uses (alpha, bias, none/ReLU, dropout, bf16 residual, bf16 gate).  It is kept small on purpose - 4 passes
    // unrolled, not 16 - because the fully unrolled general epilogue made the kernel larger than the
    // instruction cache and cost more time than the global stores it feeds.
    const bool bf16_out = d.dtype_c == JS2T_BF16;
    const int esh = bf16_out ? 1 : 2;
    const bool fast = n0 + 128 <= N && !d.preact && (d.beta == 0.f || !bf16_out) && (d.act == JS2T_ACT_NONE || d.act == JS2T_ACT_RELU) &&
                      ((((uintptr_t)d.C + ((uintptr_t)co << esh)) & 15) == 0) && ((((uintptr_t)d.ldc << esh) & 15) == 0) &&
                      (bf16_out ? ((!d.residual || ((d.ldr & 7) == 0 && (((uintptr_t)d.residual) & 15) == 0)) &&
                                   (!d.gate || ((d.ldg & 7) == 0 && (((uintptr_t)d.gate) & 15) == 0)))
                                : (!d.residual && !d.gate));
    if (fast) {
      const bool relu = d.act == JS2T_ACT_RELU, has_res = d.residual != nullptr, has_gate = d.gate != nullptr;
      const bool has_drop = d.dropout_p > 0.f;
      const bool want_ss = d.sumsq_partial != nullptr && !bf16_out;
      float ss = 0.f;
      const float keep_scale = 1.f / (1.f - d.dropout_p), res_scale = d.res_scale, gate_scale = d.gate_scale;
      const uint16_t* resp = (const uint16_t*)d.residual + n;
      const uint16_t* gatep = (const uint16_t*)d.gate + n;
#pragma unroll 1
      for (int g = 0; g < NPASS; g += 4) {
        uint4 res_pk[4], gate_pk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int m = m0 + (g + u) * 16 + r16;
          res_pk[u] = gate_pk[u] = make_uint4(0u, 0u, 0u, 0u);
          if (m < M) {
            if (has_res) res_pk[u] = *(const uint4*)(resp + (int64_t)m * d.ldr);
            if (has_gate) gate_pk[u] = *(const uint4*)(gatep + (int64_t)m * d.ldg);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ml = (g + u) * 16 + r16, m = m0 + ml;
          if (m >= M) continue;
          const f32x4_t lo = *(const f32x4_t*)(Cs + ml * 128 + (((2 * c8) ^ (ml & 7)) << 2));
          const f32x4_t hi = *(const f32x4_t*)(Cs + ml * 128 + (((2 * c8 + 1) ^ (ml & 7)) << 2));
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = v[i] * alpha + bias_r[i];
          if (relu) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          if (has_drop) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const uint32_t keep = dropout_keep4_key(drop_key, (uint32_t)(z * M + m), (uint32_t)((n >> 2) + h), d.dropout_p);
#pragma unroll
              for (int i = 0; i < 4; ++i) v[4 * h + i] = ((keep >> i) & 1u) ? v[4 * h + i] * keep_scale : 0.f;
            }
          }
          if (has_res) {
            float rr[8];
            unpack_bf16x8(res_pk[u], rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += res_scale * rr[i];
          }
          if (has_gate) {
            float rr[8];
            unpack_bf16x8(gate_pk[u], rr);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = rr[i] > 0.f ? v[i] * gate_scale : 0.f;
          }
          const int64_t coff = co + (int64_t)m * d.ldc + n;
          if (bf16_out) {
            uint4 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            pk.z = pack_bf16x2(v[4], v[5]);
            pk.w = pack_bf16x2(v[6], v[7]);
            *(uint4*)((uint16_t*)d.C + coff) = pk;
          } else {
            if (d.beta != 0.f) {  // in-place accumulation of an f32 gradient
              const float4 o0 = *(const float4*)((const float*)d.C + coff), o1 = *(const float4*)((const float*)d.C + coff + 4);
              v[0] += d.beta * o0.x, v[1] += d.beta * o0.y, v[2] += d.beta * o0.z, v[3] += d.beta * o0.w;
              v[4] += d.beta * o1.x, v[5] += d.beta * o1.y, v[6] += d.beta * o1.z, v[7] += d.beta * o1.w;
            }
            *(float4*)((float*)d.C + coff) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)((float*)d.C + coff + 4) = make_float4(v[4], v[5], v[6], v[7]);
            if (want_ss) {
#pragma unroll
              for (int i = 0; i < 8; ++i) ss = fmaf(v[i], v[i], ss);
            }
          }
        }
      }
      if (want_ss) {  // (js2t_gemm_grouped has checked that every tile of the launch takes this path)
        ss = block_sum(ss, (float*)smem);  // over the staged tile: block_sum starts with a barrier, every thread has read its rows
        if (t == 0) d.sumsq_partial[blockIdx.x] = ss;
      }
    } else {
      const uint4 none = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 1
      for (int pass = 0; pass < NPASS; ++pass) {
        const int ml = pass * 16 + r16;
        const int m = m0 + ml;
        if (m < M && n < N) {
          const f32x4_t lo = *(const f32x4_t*)(Cs + ml * 128 + (((2 * c8) ^ (ml & 7)) << 2));
          const f32x4_t hi = *(const f32x4_t*)(Cs + ml * 128 + (((2 * c8 + 1) ^ (ml & 7)) << 2));
          float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          gemm_epilogue8(d, z, co, m, n, v, alpha, drop_key, bias_r, false, none, none);
        }
      }
    }
  }
}


// ---- LayerNorm fold (js2t_gemm_desc::ln_partial / rs_partial) helpers of the register-direct epilogues
// LNF_GROUPS (gemm_shared.hpp): row length 512 = 8 groups of 64 columns
// sum over the 16 lanes of a DPP row (lanes that share lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));  // row_mirror
  return v;
}
// sum over the 4 lanes that share lane & 15 (one per DPP row)
__device__ __forceinline__ float col4_sum(float v) {
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}
__device__ __forceinline__ float bf16_round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
// Consumer side, at the start of a tile: lane L < 48 of a wave turns the eight partial {sum, sum of squares} pairs of row
// row0 + L (64 contiguous bytes, written by the producing product's eight 64-column groups) into that row's
// 1 / sqrt(var + eps) and keeps it in ONE register across the K loop; the epilogue's lane (g, e) of row block i then fetches
// the value of row 16 i + 4 g + e from lane 16 i + 4 g + e (ds_bpermute).  Requested here, not in the epilogue: a vector load
// in the epilogue of the ring kernels returns in order behind the next tile's queued LDS-DMA stages - microseconds.
__device__ __forceinline__ float lnf_row_rstd(const js2t_gemm_desc& d, int row0, int lane, bool first_col_tile) {
  const int row = row0 + lane;
  float rs = 0.f;
  if (lane < 48 && row < d.M) {
    const float4* pp = (const float4*)(d.ln_partial + (int64_t)row * (2 * LNF_GROUPS));
    const float4 a = pp[0], b = pp[1], c = pp[2], e = pp[3];
    const float s1 = ((a.x + a.z) + (b.x + b.z)) + ((c.x + c.z) + (e.x + e.z));
    const float s2 = ((a.y + a.w) + (b.y + b.w)) + ((c.y + c.w) + (e.y + e.w));
    const float inv = 1.f / (float)(64 * LNF_GROUPS), mu = s1 * inv;
    rs = 1.f / sqrtf(fmaxf(fmaf(-mu, mu, s2 * inv), 0.f) + d.ln_eps);
    if (first_col_tile && d.ln_mean) d.ln_mean[row] = mu, d.ln_rstd[row] = rs;  // for the LayerNorm backward
  }
  return rs;
}

// Register-direct epilogue for the permuted accumulator layout (k-contiguous B operand): lane (g = lane>>4, r = lane&15)
// holds, for each of its MI row blocks, the 16 consecutive columns n0 + 64*wn + 16g .. +15 of row 16i + r.  Returns false
// (nothing done) when the tile needs the general path: partial N tile, unaligned rows, pre-activation output, beta,
// activations other than ReLU, f32 residual / gate.
template <int BM>
__device__ __forceinline__ bool direct_tile_epilogue(const js2t_gemm_desc& d, f32x4_t (&acc)[BM / 32][4], int z, int64_t co, int m0,
                                                     int n0) {
  constexpr int MI = BM / 32;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int M = d.M, N = d.N;
  const bool bf16_out = d.dtype_c == JS2T_BF16;
  const int esh = 1;
  const bool fast = bf16_out && n0 + 128 <= N && !d.preact && d.beta == 0.f && !(d.residual && d.gate) && (d.act == JS2T_ACT_NONE || d.act == JS2T_ACT_RELU) &&
                    ((((uintptr_t)d.C + ((uintptr_t)co << esh)) & 15) == 0) && ((((uintptr_t)d.ldc << esh) & 15) == 0) &&
                    (bf16_out ? ((!d.residual || ((d.ldr & 7) == 0 && (((uintptr_t)d.residual) & 15) == 0)) &&
                                 (!d.gate || ((d.ldg & 7) == 0 && (((uintptr_t)d.gate) & 15) == 0)))
                              : (!d.residual && !d.gate));
  if (!fast) return false;
  const int n = n0 + wn * 64 + 16 * (lane >> 4);
  const int mrow = m0 + wm * (BM / 2) + (lane & 15);
  // LayerNorm fold (js2t_gemm has checked the preconditions).  A lane owns whole rows here (row mrow + 16 i, sixteen
  // consecutive columns), so the consumer side reads the eight partial pairs of its OWN rows - no tile follows in this
  // kernel, the loads queue behind nothing - and the producer side sums a row's 64-column group over four lanes.
  const bool lnf = d.ln_partial != nullptr, wstats = d.rs_partial != nullptr;
  float ln_rs[MI];
  if (lnf) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = min(mrow + 16 * i, M - 1);
      const float4* pp = (const float4*)(d.ln_partial + (int64_t)row * (2 * LNF_GROUPS));
      const float4 a = pp[0], b = pp[1], c = pp[2], e = pp[3];
      const float s1 = ((a.x + a.z) + (b.x + b.z)) + ((c.x + c.z) + (e.x + e.z));
      const float s2 = ((a.y + a.w) + (b.y + b.w)) + ((c.y + c.w) + (e.y + e.w));
      const float inv = 1.f / (float)(64 * LNF_GROUPS), mu = s1 * inv;
      ln_rs[i] = 1.f / sqrtf(fmaxf(fmaf(-mu, mu, s2 * inv), 0.f) + d.ln_eps);
      if (n == 0 && d.ln_mean && mrow + 16 * i < M) d.ln_mean[row] = mu, d.ln_rstd[row] = ln_rs[i];
    }
  }
  const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f);
  const bool relu = d.act == JS2T_ACT_RELU, has_res = d.residual != nullptr, has_gate = d.gate != nullptr;
  const bool has_drop = d.dropout_p > 0.f;
  const uint32_t drop_key = has_drop ? dropout_key(d.rng_state, d.rng_stream) : 0u;
  const float keep_scale = 1.f / (1.f - d.dropout_p), res_scale = d.res_scale, gate_scale = d.gate_scale;
  // residual / gate rows first: their latency overlaps the bias fetch and the arithmetic of the earlier rows
  uint4 rg[MI][2];
  if (has_res || has_gate) {
    const uint16_t* src = (const uint16_t*)(has_res ? d.residual : d.gate) + n;
    const int64_t ld = has_res ? d.ldr : d.ldg;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = min(mrow + 16 * i, M - 1);
      rg[i][0] = *(const uint4*)(src + (int64_t)m * ld);
      rg[i][1] = *(const uint4*)(src + (int64_t)m * ld + 8);
    }
  }
  float bias_r[16];
  if (d.bias && (((uintptr_t)d.bias) & 15) == 0) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float4 b4 = *(const float4*)(d.bias + n + 4 * h);
      bias_r[4 * h] = b4.x, bias_r[4 * h + 1] = b4.y, bias_r[4 * h + 2] = b4.z, bias_r[4 * h + 3] = b4.w;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) bias_r[c] = d.bias ? d.bias[n + c] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow + 16 * i;
    float v[16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[4 * j + r] = lnf ? fmaf(ln_rs[i], acc[i][j][r], bias_r[4 * j + r]) : acc[i][j][r] * alpha + bias_r[4 * j + r];
    if (relu) {
#pragma unroll
      for (int c = 0; c < 16; ++c) v[c] = fmaxf(v[c], 0.f);
    }
    if (has_drop) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const uint32_t keep = dropout_keep4_key(drop_key, (uint32_t)(z * M + m), (uint32_t)((n >> 2) + h), d.dropout_p);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[4 * h + c] = ((keep >> c) & 1u) ? v[4 * h + c] * keep_scale : 0.f;
      }
    }
    if (has_res) {
      float rr[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unpack_bf16x8(rg[i][h], rr);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[8 * h + c] += res_scale * rr[c];
      }
    }
    if (has_gate) {
      float rr[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unpack_bf16x8(rg[i][h], rr);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[8 * h + c] = rr[c] > 0.f ? v[8 * h + c] * gate_scale : 0.f;
      }
    }
    if (d.dot_partial) {  // sum over this lane's 16 columns of bf16(C) * dot_src, then over the 64-column group's four lanes
      const int mm = min(m, M - 1);
      const uint16_t* dsrc = (const uint16_t*)d.dot_src + (int64_t)mm * d.ld_dot + n;
      float rr[8], sd = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        unpack_bf16x8(*(const uint4*)(dsrc + 8 * h), rr);
#pragma unroll
        for (int c = 0; c < 8; ++c) sd = fmaf(bf16_round(v[8 * h + c]), rr[c], sd);
      }
      sd = col4_sum(sd);
      if ((lane >> 4) == 0 && m < M) d.dot_partial[(int64_t)m * (N >> 6) + ((n0 >> 6) + wn)] = sd;
    }
    if (wstats) {  // sums of what is stored (bf16-rounded) over this lane's 16 columns, then over the group's four lanes
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float q = bf16_round(v[c]);
        s1 += q;
        s2 = fmaf(q, q, s2);
      }
      s1 = col4_sum(s1), s2 = col4_sum(s2);
      if ((lane >> 4) == 0 && m < M)
        *(float2*)(d.rs_partial + 2 * ((int64_t)m * LNF_GROUPS + ((n0 >> 6) + wn))) = make_float2(s1, s2);
    }
    if (m < M) {
      const int64_t coff = co + (int64_t)m * d.ldc + n;
      if (bf16_out) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint4 pk;
          pk.x = pack_bf16x2(v[8 * h + 0], v[8 * h + 1]);
          pk.y = pack_bf16x2(v[8 * h + 2], v[8 * h + 3]);
          pk.z = pack_bf16x2(v[8 * h + 4], v[8 * h + 5]);
          pk.w = pack_bf16x2(v[8 * h + 6], v[8 * h + 7]);
          *(uint4*)((uint16_t*)d.C + coff + 8 * h) = pk;
        }
      } else {
#pragma unroll
        for (int h = 0; h < 4; ++h)
          *(float4*)((float*)d.C + coff + 4 * h) = make_float4(v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
      }
    }
  }
  return true;
}

// BM = 128 (default) or 64: the 64-row tile doubles the number of blocks for outputs with few tiles (N = 512 layers,
// decoder-sized M) at the price of re-reading the B panel twice as often; transposed A images are 128 rows only.
template <int BM, bool TA, bool TB, bool SPLITK, int NST = 2>
__device__ __forceinline__ void dma_gemm_block(const js2t_gemm_desc& d, int tiles_m, int tiles_n, float* c_atomic, int split_k,
                                               int z, int lid, int slice) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(BM == 128 || (BM == 64 && !TA), "64-row tiles need a k-contiguous A operand");
  constexpr int A_TILE = TA ? 16384 : BM * 128, TILE = 16384, STAGE = A_TILE + TILE;
  constexpr int NPA = TA ? 4 : BM / 32, MI = BM / 32;  // DMA pieces per wave for A; 16-row MFMA blocks per wave
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * F_BN;
  int64_t ao, bo, co;
  batch_offsets(d, z, ao, bo, co);
  const uint16_t* Ab = (const uint16_t*)d.A + ao;
  const uint16_t* Bb = (const uint16_t*)d.B + bo;
  const int wm = w >> 1, wn = w & 1;
  const int M = d.M, N = d.N, K = d.K;
  const int64_t lda = d.lda, ldb = d.ldb;
  f32x4_t acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (K + F_BK - 1) / F_BK;
  const int per = SPLITK ? (nk_all + split_k - 1) / split_k : nk_all;
  const int kt0 = SPLITK ? slice * per : 0;
  const int nk = min(per, nk_all - kt0);
  if (SPLITK && nk <= 0) return;

  // optional row sums of op(A) (bias gradient of a weight-gradient product): the first tile column's wn == 0 waves
  // multiply their A fragments with an all-ones fragment, 4 extra MFMAs per 32 k
  const bool do_rs = TA && TB && d.a_rowsum != nullptr && n0 == 0 && wn == 0;
  f32x4_t racc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) racc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8_ones_t;
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, s16x8_ones_t{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});

  // NST stages of 64 k.  Two by default: with the chip full (>= 2-3 blocks per CU) the other resident blocks hide the DMA
  // latency, and a third stage on the 64-row tile (72 KB -> two blocks per CU instead of three) made those products 25-60 %
  // slower.  Small grids (decoder-sized M: fewer blocks than CUs) have nothing else resident and LDS to spare: there
  // NST = 4 keeps three stages in flight (counted vmcnt, raw s_barrier - __syncthreads() would drain the DMA queue).
  constexpr int PER = NPA + 4;  // DMA instructions per wave and stage
  // per-lane source offsets of a stage = offsets at k = 0 (computed once) + a wave-uniform step; only a partial last
  // stage needs the clamped form (recomputing row * ld per stage cost 24 quarter-rate multiplies per K step on the
  // reduction-major operands, where the clamp on k keeps the compiler from hoisting them)
  int64_t oa0[NPA], ob0[4], oa[NPA], ob[4];
  dma_offsets<TA, NPA>(lda, m0, M, 0, 0x7fffffff, t, oa0);
  dma_offsets<TB, 4, !TB>(ldb, n0, N, 0, 0x7fffffff, t, ob0);
  auto issue_stage = [&](int s, int slot) {
    const int k0 = (kt0 + s) * F_BK;
    if (k0 + F_BK <= K) {
      const int64_t sa = TA ? (int64_t)k0 * lda : (int64_t)k0, sb = TB ? (int64_t)k0 * ldb : (int64_t)k0;
#pragma unroll
      for (int q = 0; q < NPA; ++q) oa[q] = oa0[q] + sa;
#pragma unroll
      for (int q = 0; q < 4; ++q) ob[q] = ob0[q] + sb;
    } else {
      dma_offsets<TA, NPA>(lda, m0, M, k0, K, t, oa);
      dma_offsets<TB, 4, !TB>(ldb, n0, N, k0, K, t, ob);
    }
    dma_issue<NPA, TA || TB>(Ab, oa, smem + slot * STAGE, t);
    dma_issue<4, TA || TB>(Bb, ob, smem + slot * STAGE + A_TILE, t);
  };
  issue_stage(0, 0);
#pragma unroll
  for (int st = 1; st < NST - 1; ++st)
    if (st < nk) issue_stage(st, st);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = (kt0 + kt) * F_BK;
    // stage kt has been issued; wait for it, patch a partial K tail, make it visible to all waves
    if (NST >= 4 && kt + 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    } else if (NST >= 3 && kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (k0 + F_BK > K) {
      dma_zero_tail<TA, NPA>(smem + cur * STAGE, K - k0, t);
      dma_zero_tail<TB, 4, !TB>(smem + cur * STAGE + A_TILE, K - k0, t);
    }
    if (NST == 2) {
      __syncthreads();
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // the slot requested now was consumed in iteration kt-1, which every wave has left (barrier above)
    if (kt + NST - 1 < nk) issue_stage(kt + NST - 1, cur == 0 ? NST - 1 : cur - 1);
    const unsigned char* At = smem + cur * STAGE;
    const unsigned char* Bt = At + A_TILE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t fn[4], fm[MI];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (!TB) {
          // fragment row rho of accumulator j is tile row 16*(rho>>2) + 4j + (rho&3): a lane (group g) then owns the 16
          // CONSECUTIVE output columns 16g .. 16g+15 of its rows, i.e. 32-byte bf16 runs it can store straight
          // from registers (no LDS round trip, no barriers in the epilogue)
          const int row = wn * 64 + ((lane & 15) >> 2) * 16 + j * 4 + (lane & 3), c = kk * 4 + (lane >> 4);
          fn[j] = *(const bf16x8_t*)(Bt + row * 128 + ((c ^ kc_key<true>(row)) << 4));
        } else {
          // reduction-major B image: the same column ownership costs a 2-way bank conflict on the transposing reads
          // (the 8-byte offset inside a 32-byte group would be the same for all lanes) - measured slower, so these
          // variants keep the natural fragment order and the LDS-staged epilogue
          fn[j] = frag_load2<TB>(Bt, wn * 64 + 16 * j, kk, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) fm[i] = frag_load2<TA>(At, wm * (BM / 2) + 16 * i, kk, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[j], fm[i], acc[i][j], 0, 0, 0);
      if (do_rs) {
#pragma unroll
        for (int i = 0; i < MI; ++i) racc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm[i], racc[i], 0, 0, 0);
      }
    }
    cur = cur + 1 == NST ? 0 : cur + 1;
  }
  if (do_rs && (lane >> 4) == 0) {  // every output row of the ones-product holds the same sums: take row 0
    typedef __attribute__((address_space(1))) float gfloat;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * (BM / 2) + 16 * i + lane;
      if (m < M) __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)d.a_rowsum + m, racc[i][0]);
    }
  }
  if constexpr (!SPLITK && !TB) {
    if (direct_tile_epilogue<BM>(d, acc, z, co, m0, n0)) return;
  }
  dma_tile_epilogue<BM, SPLITK, !TB>(d, acc, smem, z, co, m0, n0, c_atomic);
}

template <int BM, bool TA, bool TB, bool SPLITK, int NST = 2>
__global__ __launch_bounds__(256, 2) void gemm_bf16_dma_kernel(js2t_gemm_desc d, int tiles_m, int tiles_n, float* c_atomic,
                                                               int split_k) {
  dma_gemm_block<BM, TA, TB, SPLITK, NST>(d, tiles_m, tiles_n, c_atomic, split_k, blockIdx.y, xcd_remap(blockIdx.x, tiles_m * tiles_n),
                                          blockIdx.z);
}

// Grouped launch: up to JS2T_GEMM_GROUP_MAX independent products of identical shape whose operands live at unrelated
// addresses (the deferred weight gradients of one layer type: see js2t_gemm_grouped).  1-D grid over (member, K slice,
// tile), XCD-aware as a whole: an XCD gets whole members (16 members: two each), so each member's dY and X stream
// through ONE L2 - with the tiles of every member spread over all eight XCDs (one remap per member) each XCD read
// every member's X: 2.4 GB instead of 1 GB per launch of the FFN weight gradients, at fabric speed.
struct GemmGroup {
  const void* A[JS2T_GEMM_GROUP_MAX];
  const void* B[JS2T_GEMM_GROUP_MAX];
  void* C[JS2T_GEMM_GROUP_MAX];
  float* rowsum[JS2T_GEMM_GROUP_MAX];
};
template <int BM, bool TA, bool TB, bool SPLITK>
__global__ __launch_bounds__(256, 2) void gemm_bf16_dma_grouped_kernel(js2t_gemm_desc d, GemmGroup grp, int tiles_m, int tiles_n,
                                                                       int split_k) {
  const int tiles = tiles_m * tiles_n, nsplit = SPLITK ? split_k : 1;
  const int lid_all = xcd_remap(blockIdx.x, gridDim.x);
  const int vm = lid_all / tiles, tile = lid_all - vm * tiles;  // vm = member * nsplit + slice
  const int g = vm / nsplit, slice = vm - g * nsplit;
  js2t_gemm_desc dd = d;
  dd.A = grp.A[g];
  dd.B = grp.B[g];
  dd.C = grp.C[g];
  dd.a_rowsum = grp.rowsum[g];
  dma_gemm_block<BM, TA, TB, SPLITK>(dd, tiles_m, tiles_n, (float*)dd.C, split_k, 0, tile, slice);
}

__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};
// ------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves, half-tile DMA ring: large k-contiguous products (A [M,K], B [N,K], bf16 C)
// ------------------------------------------------------------------------------------------------
// The 128x128 kernels above top out near 1000 TFLOP/s because every 2.1 MFLOP of a K step pull 32 KB from L2 into LDS
// (about 16 TB/s aggregate at that rate).  A 256x256 tile halves the bytes per flop; with one 512-thread block per CU
// nothing else is resident to hide latency, so the pipeline is explicit:
//  * a K tile (64 k) arrives as four 16 KB half-tiles - A0 (rows 0-127), B0 (cols 0-127), B1, A1 - each its own DMA
//    unit (2 global_load_lds per wave); two K tiles of slots (8 x 16 KB = 128 KB) form a ring, the half-tile of the NEXT
//    K tile is requested in the phase that first uses its counterpart, i.e. four halves ahead;
//  * a K tile is multiplied in four phases of 16 MFMAs per wave, one quadrant (A half x B half) each, ordered
//    (A0,B0) (A0,B1) (A1,B1) (A1,B0) so that every phase needs at most one half-tile that the previous one did not;
//    each phase waits with a COUNTED vmcnt for exactly that half-tile (later ones stay in flight) and crosses one raw
//    s_barrier (a __syncthreads() fence would drain the DMA queue);
//  * wave (wr, wc) owns rows {64wr..64wr+63} of both A halves and columns {32wc..32wc+31} of both B halves, with the
//    B fragment rows permuted so that a lane ends up with 8 consecutive output columns -> 16-byte stores from registers.
constexpr int W_HALF = 16384;
constexpr int W_LDS = 8 * W_HALF;

__device__ __forceinline__ bf16x8_t w256_frag(const unsigned char* img, int row, int kk, int lane) {
  const int c = kk * 4 + (lane >> 4);
  return *(const bf16x8_t*)(img + row * 128 + ((c ^ (row & 7)) << 4));
}

__global__ __launch_bounds__(512, 1) void gemm_bf16_w256_kernel(js2t_gemm_desc d, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4;
  const int wr = w >> 2, wc = w & 3;
  const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * 256, n0 = (lid % tiles_n) * 256;
  const int M = d.M, N = d.N, K = d.K;
  const int nk = (K + 63) >> 6;
  const int H = 4 * nk;  // half-tiles in the stream

  // DMA sources: stream kind 0 = A0, 1 = B0, 2 = B1, 3 = A1; two 1 KB pieces per wave and half-tile
  const uint16_t* src[4][2];
  int kc;
  {
    const uint16_t* Ab = (const uint16_t*)d.A;
    const uint16_t* Bb = (const uint16_t*)d.B;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = (w * 2 + q) * 64 + lane;
      const int row = p >> 3, slot = p & 7, chunk = slot ^ (row & 7);
      src[0][q] = Ab + (int64_t)min(m0 + row, M - 1) * d.lda + chunk * 8;
      src[3][q] = Ab + (int64_t)min(m0 + 128 + row, M - 1) * d.lda + chunk * 8;
      src[1][q] = Bb + (int64_t)min(n0 + row, N - 1) * d.ldb + chunk * 8;
      src[2][q] = Bb + (int64_t)min(n0 + 128 + row, N - 1) * d.ldb + chunk * 8;
    }
    kc = (((lane & 7) ^ ((((w * 2) * 64 + lane) >> 3) & 7)) << 3);
  }
  auto issue_half = [&](int h) {  // h = global half-tile index
    const int kt = h >> 2, kind = h & 3;
    unsigned char* img = smem + ((kt & 1) * 4 + kind) * W_HALF;
    const bool live = kt * 64 + kc < K;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint16_t* sp = kind == 0 ? src[0][q] : kind == 1 ? src[1][q] : kind == 2 ? src[2][q] : src[3][q];
      const uint16_t* gsrc = live ? sp + kt * 64 : (const uint16_t*)&g_zero16;
      __builtin_amdgcn_global_load_lds((g_cvoid*)gsrc, (l_void*)(img + (w * 2 + q) * 1024), 16, 0, 0);
    }
  };
  // wait until at most `behind` younger half-tiles of this wave's own DMA are in flight
  auto wait_halves = [&](int behind) {
    if (behind >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (behind == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4_t acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int h = 0; h < 4; ++h) issue_half(h);  // K tile 0 (H >= 4 always)

  const int arow = wr * 64 + (lane & 15);                                        // + 16 i
  const int brow = wc * 32 + ((lane & 15) >> 2) * 8 + (lane & 3);                // + 4 j
  bf16x8_t fA[4][2], fB0[2][2], fB1[2][2];
  // one phase: wait for half-tile `need` (global index), barrier, request half-tile P + 4, read fragments, 16 MFMAs
  auto phase_sync = [&](int P, int need) {
    wait_halves(min(P + 3, H - 1) - need);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (P + 4 < H) issue_half(P + 4);
  };
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* base = smem + (kt & 1) * 4 * W_HALF;
    const unsigned char* iA0 = base, * iB0 = base + W_HALF, * iB1 = base + 2 * W_HALF, * iA1 = base + 3 * W_HALF;
    const int P = 4 * kt;
    // ---- phase 1: (A0, B0)
    phase_sync(P, P + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) fB0[j][kk] = w256_frag(iB0, brow + 4 * j, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) fA[i][kk] = w256_frag(iA0, arow + 16 * i, kk, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[0][0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fB0[j][kk], fA[i][kk], acc[0][0][i][j], 0, 0, 0);
    // ---- phase 2: (A0, B1)
    phase_sync(P + 1, P + 2);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) fB1[j][kk] = w256_frag(iB1, brow + 4 * j, kk, lane);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[0][1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fB1[j][kk], fA[i][kk], acc[0][1][i][j], 0, 0, 0);
    // ---- phase 3: (A1, B1)
    phase_sync(P + 2, P + 3);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) fA[i][kk] = w256_frag(iA1, arow + 16 * i, kk, lane);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[1][1][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fB1[j][kk], fA[i][kk], acc[1][1][i][j], 0, 0, 0);
    // ---- phase 4: (A1, B0) - fragments already in registers
    phase_sync(P + 3, P + 3);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[1][0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fB0[j][kk], fA[i][kk], acc[1][0][i][j], 0, 0, 0);
  }

  // ---- epilogue from registers: lane (g, r = lane & 15) holds 8 consecutive columns of row 16i + r of each quadrant
  const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f);
  const bool relu = d.act == JS2T_ACT_RELU, has_res = d.residual != nullptr, has_gate = d.gate != nullptr;
  const bool has_drop = d.dropout_p > 0.f;
  const uint32_t drop_key = has_drop ? dropout_key(d.rng_state, d.rng_stream) : 0u;
  const float keep_scale = 1.f / (1.f - d.dropout_p), res_scale = d.res_scale, gate_scale = d.gate_scale;
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    const int n = n0 + hb * 128 + wc * 32 + g * 8;
    if (n >= N) continue;  // N is a multiple of 8 (dispatcher): a column group is in or out as a whole
    float bias_r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) bias_r[c] = d.bias ? d.bias[n + c] : 0.f;
#pragma unroll
    for (int ha = 0; ha < 2; ++ha) {
      uint4 rg[4];
      if (has_res || has_gate) {
        const uint16_t* sp = (const uint16_t*)(has_res ? d.residual : d.gate) + n;
        const int64_t ld = has_res ? d.ldr : d.ldg;
#pragma unroll
        for (int i = 0; i < 4; ++i) rg[i] = *(const uint4*)(sp + (int64_t)min(m0 + ha * 128 + wr * 64 + 16 * i + (lane & 15), M - 1) * ld);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + ha * 128 + wr * 64 + 16 * i + (lane & 15);
        float v[8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * j + r] = acc[ha][hb][i][j][r] * alpha + bias_r[4 * j + r];
        if (relu) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
        }
        if (has_drop) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const uint32_t keep = dropout_keep4_key(drop_key, (uint32_t)m, (uint32_t)((n >> 2) + h2), d.dropout_p);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * h2 + c] = ((keep >> c) & 1u) ? v[4 * h2 + c] * keep_scale : 0.f;
          }
        }
        if (has_res) {
          float rr[8];
          unpack_bf16x8(rg[i], rr);
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] += res_scale * rr[c];
        }
        if (has_gate) {
          float rr[8];
          unpack_bf16x8(rg[i], rr);
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = rr[c] > 0.f ? v[c] * gate_scale : 0.f;
        }
        if (m < M) {
          uint4 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          pk.z = pack_bf16x2(v[4], v[5]);
          pk.w = pack_bf16x2(v[6], v[7]);
          *(uint4*)((uint16_t*)d.C + (int64_t)m * d.ldc + n) = pk;
        }
      }
    }
  }
}

// products this kernel takes: k-contiguous bf16 operands, bf16 C, the epilogue terms of direct_tile_epilogue
// ... and that are big enough for it to pay: one 512-thread block per CU has no second block to hide its prologue and
// epilogue behind, and at fewer than two full rounds of 256x256 tiles the quantisation loss eats the gain (measured:
// 8192^3 1069 vs 960 TFLOP/s for the 128x128 kernel, but FFN1 12000x2048x512 47 vs 39 us)
inline bool w256_eligible(const js2t_gemm_desc& d) {
  if (d.ln_partial || d.rs_partial) return false;  // the LayerNorm fold lives in the persistent 192x128 kernels only
  if (d.trans_a || d.trans_b || d.conv || d.split_k > 1 || d.batch != 1 || d.dtype_c != JS2T_BF16) return false;
  if ((d.N & 7) || d.M < 256 || d.N < 256) return false;
  if (!g_force_w256 && ((int64_t)cdiv(d.M, 256) * cdiv(d.N, 256) < 512 || d.K < 1024)) return false;
  if (d.preact || d.beta != 0.f || !(d.act == JS2T_ACT_NONE || d.act == JS2T_ACT_RELU) || (d.residual && d.gate)) return false;
  if ((((uintptr_t)d.C) & 15) || (d.ldc & 7)) return false;
  if (d.residual && ((d.ldr & 7) || (((uintptr_t)d.residual) & 15))) return false;
  if (d.gate && ((d.ldg & 7) || (((uintptr_t)d.gate) & 15))) return false;
  return true;
}
int launch_bf16_w256(const js2t_gemm_desc& d, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_w256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tm = cdiv(d.M, 256), tn = cdiv(d.N, 256);
  hipLaunchKernelGGL(gemm_bf16_w256_kernel, dim3(tm * tn), dim3(512), W_LDS, s, d, tm, tn);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// ------------------------------------------------------------------------------------------------
// 192x128x64 tile, 4 waves of 96x64, persistent blocks with one DMA ring that runs across tiles
// ------------------------------------------------------------------------------------------------
// The 128x128 / 64x128 kernels above pull 32 / 24.6 KB from L2 into LDS per 2.1 / 1.05 MFLOP K step and sit at the
// L2 -> LDS delivery rate (about 16-19 TB/s aggregate; MI355X_MICROARCH.md 'gather into LDS') long before the MFMA
// pipe is busy.  This tile moves 40 KB per 3.1 MFLOP (0.55x the bytes per flop of the 64-row tile) and its 96x64 wave
// tile needs 0.42 of the LDS read bandwidth at full MFMA rate.  One 256-thread block per CU has nothing else resident
// to hide latency behind, so:
//  * the grid is one block per CU; block b walks tiles b, b + G, ... (XCD-aware order) and treats all their K steps as
//    ONE stream of 40 KB stages through a 3-slot ring: stages s+2 and s+3 are in flight while stage s is multiplied,
//    also across a tile boundary, so a tile's epilogue overlaps the next tile's first loads;
//  * one raw s_barrier per stage, in the MIDDLE of it: by then every wave holds the second k-half of stage s in
//    registers, so slot s % 3 is free for stage s + 3, and a counted vmcnt has retired this wave's part of stage s + 1;
//  * fragments are double-buffered by k-half: the reads of the next half are issued before the 24 MFMAs of the current
//    one;
//  * B fragment rows are permuted as in dma_gemm_block, so each lane owns 16 consecutive output columns and the tile is
//    stored from registers (direct_tile_epilogue).
constexpr int P_BM = 192, P_ATILE = P_BM * 128, P_STAGE = P_ATILE + 16384, P_NST = 3, P_LDS = P_NST * P_STAGE, P_PER = 10;
// ring depth of the loader / consumer form: 3 slots (120 KB); -DJS2T_P192S_NST4: 4 slots = the CU's whole 160 KB (measured, see profiles/README.md)
#ifdef JS2T_P192S_NST4
constexpr int PS_NST = 4;
#else
constexpr int PS_NST = 3;
#endif
constexpr int PS_LDS = PS_NST * P_STAGE;
// NST = 2 (js2t_gemm_p192_ring(2)): a two-slot ring is 80 KB, so TWO blocks fit a CU and the grid is two blocks per CU.  Each
// block then has one stage in flight instead of two (its ten requests go out in the second k-half of a stage and must have
// landed by the middle of the next one) - what one block cannot hide any more (request latency, its epilogue, the switch
// to the next tile, the prologue of the launch) is the other block's compute time.
#ifdef JS2T_P192_PROF
__device__ unsigned long long g_p192_prof2[8];
#define P192_E(i)                                                   \
  do {                                                              \
    const unsigned long long c_ = __builtin_readcyclecounter();     \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_p192_prof2[i] += c_ - e_last_; \
    e_last_ = c_;                                                   \
  } while (0)
__device__ unsigned long long g_p192_prof[8];
#define P192_T(i)                                          \
  do {                                                     \
    const unsigned long long c_ = __builtin_readcyclecounter(); \
    prof_[i] += c_ - last_;                                \
    last_ = c_;                                            \
  } while (0)
#else
#define P192_T(i)
#define P192_E(i)
#endif

typedef int frag_i4 __attribute__((ext_vector_type(4)));  // a bf16x8 fragment carried across loop iterations as 4 dwords
__device__ __forceinline__ bf16x8_t as_bf16x8(const frag_i4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// 16-byte granule index of B element block (tile column n, k-chunk c) inside a stage's B image.  Fragment j of a lane
// (r = lane & 15) is column 8r + j, so that after eight MFMAs along N a lane holds 8 CONSECUTIVE output columns of each
// of its rows and the 16 lanes of a row group store one contiguous 256-byte run.  The image keeps whole 128-byte
// source lines inside a DMA piece (8 lines per instruction) and is conflict-free for the ds_read_b128 lane groups.
__device__ __forceinline__ int p192_b_granule(int n, int c) {
  const int r = n >> 3, j = n & 7;
  return j * 128 + (r >> 3) * 64 + (r & 7) * 8 + (c ^ (r >> 1));
}

// epilogue of one wave's 48 x 128 sub-tile from registers: lane (g, r) holds, for row block i and register e, row
// 16i + 4g + e, columns 8r .. 8r+7
// EPI < 0: every epilogue term is decided at run time; EPI >= 0: a bit mask of the terms that are present (alpha = 1),
// so that the variants the train step uses carry no dead branches - with one wave per SIMD nothing overlaps the
// epilogue, its instruction count is paid in full.
template <int EPI>
__device__ __forceinline__ void p192_load_bias(const js2t_gemm_desc& d, int n, float (&bias_r)[8]) {
  const bool has_bias = (EPI < 0 ? d.bias != nullptr : (EPI & PE_BIAS) != 0) && n < d.N;  // n >= N: a column group of the N tail
  if (has_bias && (((uintptr_t)d.bias) & 15) == 0) {
    const float4 b0 = *(const float4*)(d.bias + n), b1 = *(const float4*)(d.bias + n + 4);
    bias_r[0] = b0.x, bias_r[1] = b0.y, bias_r[2] = b0.z, bias_r[3] = b0.w;
    bias_r[4] = b1.x, bias_r[5] = b1.y, bias_r[6] = b1.z, bias_r[7] = b1.w;
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) bias_r[c] = has_bias ? d.bias[n + c] : 0.f;
  }
}
// Round 6: the bias + ReLU + dropout (+ folded LayerNorm) epilogue of FFN layer 1 at half the vector instructions (81 -> 45 per row
// of eight outputs in the ISA; tools/isa_mix.py), same bits.  With one or two waves per SIMD nothing covers a tile's epilogue, its
// instruction count is paid in full.  What changed, term by term:
//  * ReLU and the dropout decision on the PACKED bf16 pair: a dropped half gets its sign bit set ((h ^ 0x8000) -sat- (thr ^ 0x8000)
//    is negative exactly when the 16-bit hash half is below the threshold: v_pk_sub_i16 clamp), then one v_pk_max_i16 against 0
//    zeroes dropped and negative halves alike.  ReLU commutes with the positive keep-scale and with rounding, so the stored value is
//    the old one (a negative zero now becomes a positive one);
//  * the row keys hash32(row ^ key): the 16 lanes of a row group share their twelve rows, so lane r computes ONE key (row
//    16 (r >> 2) + 4 g + (r & 3)) and the others fetch it by ds_swizzle (row16_bcast: an immediate pattern, no address register) -
//    the LDS crossbar is idle in the epilogue; every lane used to hash all twelve;
//  * the row's address: a wave-uniform row base (scalar unit) + one lane offset for the whole tile.
// Measured and NOT taken: rstd * acc + bias and the keep-scale as v_pk_fma_f32 / v_pk_mul_f32 over the two rows an accumulator pair
// holds - slower (30.7 against 29.7 us; the guide's 'packed f32 is an anti-lever') and WRONG on the hardware: the compiler reads the
// odd bias registers through op_sel straight out of a global_load_dwordx4's return, and in ~1 of 10^5 rows the low half came out
// with the previous tile's bias in lanes 48-63 (tools/epi_diag.py; 60 wait states in front changed nothing; scalar f32: clean).
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_drop_bf16x2(uint32_t packed, uint32_t h, uint32_t thr_b) {
  const i16x2_t dd = __builtin_elementwise_sub_sat(__builtin_bit_cast(i16x2_t, h ^ 0x80008000u), __builtin_bit_cast(i16x2_t, thr_b));
  const uint32_t x = (__builtin_bit_cast(uint32_t, dd) & 0x80008000u) | packed;
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, x), i16x2_t{0, 0}));
}
// x of lane 16 (lane / 16) + k, for every lane: ds_swizzle in bit-mask mode (new lane = (lane & 0x10) | k inside each half wave).
// k is a constant at every call site after unrolling (the pattern is an immediate).
__device__ __forceinline__ uint32_t row16_bcast(uint32_t x, int k) {
#define JS2T_SWZ(K) case K: return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x10 | (K << 5));
  switch (k) {
    JS2T_SWZ(0) JS2T_SWZ(1) JS2T_SWZ(2) JS2T_SWZ(3) JS2T_SWZ(4) JS2T_SWZ(5) JS2T_SWZ(6) JS2T_SWZ(7)
    JS2T_SWZ(8) JS2T_SWZ(9) JS2T_SWZ(10) JS2T_SWZ(11) JS2T_SWZ(12) JS2T_SWZ(13) JS2T_SWZ(14)
    default: return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x10 | (15 << 5));
  }
#undef JS2T_SWZ
}
// the key of the row this lane computes for its row group (lane r -> row 16 (r >> 2) + 4 g + (r & 3) of the wave's 48; r >= 12: unused)
__device__ __forceinline__ uint32_t row16_own_key(int mw, int lane, uint32_t drop_key) {
  const int g = lane >> 4, r = lane & 15;
  return hash32((uint32_t)(mw + 16 * (r >> 2) + 4 * g + (r & 3)) ^ drop_key);
}
template <bool LNF>
__device__ __forceinline__ void p192_store_tile_ffn1(const js2t_gemm_desc& d, f32x4_t (&acc)[3][8], int mw, int n0, int lane,
                                                     const float (&bias_r)[8], uint32_t drop_key, float my_rs) {
  const int g = lane >> 4, r = lane & 15;
  const int M = d.M, n = n0 + 8 * r;
  if (n >= d.N) return;
  const float keep_scale = 1.f / (1.f - d.dropout_p);
  const uint32_t thr = (uint32_t)(d.dropout_p * 65536.0f);
  const uint32_t thr_b = ((thr ^ 0x8000u) & 0xffffu) * 0x10001u;  // (thr - 32768) in both halves: the signed twin of the unsigned compare
  const uint32_t my_key = row16_own_key(mw, lane, drop_key);
  const uint32_t colk = 2u * (uint32_t)(n >> 2);
  uint16_t* const cl = (uint16_t*)d.C + (4 * g * d.ldc + n);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[8];
      if (LNF) {
        const float rs = __shfl(my_rs, 16 * i + 4 * g + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(rs, acc[i][j][e], bias_r[j]) * keep_scale;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (acc[i][j][e] + bias_r[j]) * keep_scale;
      }
      const uint32_t rowkey = row16_bcast(my_key, 4 * i + e) + colk;
      uint4 pk;
      uint32_t* pw = (uint32_t*)&pk;
#pragma unroll
      for (int q = 0; q < 4; ++q) pw[q] = relu_drop_bf16x2(pack_bf16x2(v[2 * q], v[2 * q + 1]), hash32w(rowkey + (uint32_t)q), thr_b);
      const int mu = mw + 16 * i + e;  // wave-uniform part of the row
      if (mu + 4 * g < M) *(uint4*)(cl + (int64_t)mu * d.ldc) = pk;
    }
  }
}

// bias_r: the lane's 8 bias values, fetched when the tile started; drop_key: fetched when the kernel started (both
// would otherwise expose a dependent global-load latency per tile)
// OUT8 (the e4m3 kernels only): d.c8 != NULL adds a second output, the result as e4m3 bytes with the delayed scale of d.c8_state
// (the operand of a FOLLOWING e4m3 product, quantised here instead of by a pass of its own); d.C may then be NULL.
template <int EPI, bool OUT8 = false>
__device__ __forceinline__ void p192_store_tile(const js2t_gemm_desc& d, f32x4_t (&acc)[3][8], int mw, int n0, int lane,
                                                const float (&bias_r)[8], uint32_t drop_key, float my_rs = 0.f) {
#ifdef JS2T_GEMM_NOEPI  // measurement only (tools/k512_ceiling.py): ring + MFMAs, the tile is dropped - what the feed alone allows
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
  return;
#endif
#ifndef JS2T_OLD_FFN1_EPI
  if constexpr (EPI >= 0 && !OUT8 && (EPI & ~PE_LNF) == (PE_BIAS | PE_RELU | PE_DROP)) {
    if (n0 + 128 <= d.N) {  // whole column tile (wave-uniform): every lane of a row group is there to hand its row key over
      p192_store_tile_ffn1<(EPI & PE_LNF) != 0>(d, acc, mw, n0, lane, bias_r, drop_key, my_rs);
      return;
    }
  }
#endif
  const int g = lane >> 4, r = lane & 15;
  const int M = d.M, n = n0 + 8 * r;
  if (n >= d.N) return;  // N is a multiple of 8: a lane's column group lies inside or outside as a whole
  const float alpha = EPI < 0 ? d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f) : 1.f;
  const bool has_bias = EPI < 0 ? d.bias != nullptr : (EPI & PE_BIAS) != 0;
  const bool relu = EPI < 0 ? d.act == JS2T_ACT_RELU : (EPI & PE_RELU) != 0;
  const bool has_res = EPI < 0 ? d.residual != nullptr : (EPI & PE_RES) != 0;
  const bool has_gate = EPI < 0 ? d.gate != nullptr : (EPI & PE_GATE) != 0;
  const bool has_drop = EPI < 0 ? d.dropout_p > 0.f : (EPI & PE_DROP) != 0;
  const float keep_scale = 1.f / (1.f - d.dropout_p), res_scale = d.res_scale, gate_scale = d.gate_scale;
  const uint32_t thr = (uint32_t)(d.dropout_p * 65536.0f);
  const uint16_t* rsrc = (const uint16_t*)(has_res ? d.residual : d.gate) + n;
  const int64_t rld = has_res ? d.ldr : d.ldg;
  // LayerNorm fold, consumer side: v = rstd(m) * acc + bias (the weights are gamma-scaled and row-centred, js2t_fold_ln_weights);
  // specialised instantiations only (launch_bf16_p192 refuses other combinations)
  constexpr bool lnf = EPI >= 0 && (EPI & PE_LNF) != 0;
  float q8_inv = 0.f, q8_max = 0.f;
  if constexpr (OUT8) {
    if (d.c8) {
      const float S = d.c8_state[0];
      q8_inv = S > 0.f ? 1.f / S : 0.f;
    }
  }
  // one row key per lane, shared by row16_bcast (round 6) - in whole column tiles only: a lane outside N has returned above and
  // cannot hand its key over
  const bool share_keys = n0 + 128 <= d.N;
  const uint32_t my_key = (has_drop && share_keys) ? row16_own_key(mw, lane, drop_key) : 0u;
  // residual / gate rows: block i + 1 is requested before block i is used (twelve rows at once cost too many registers)
  uint4 rg[3][4];
  auto load_rg = [&](int i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) rg[i][e] = *(const uint4*)(rsrc + (int64_t)min(mw + 16 * i + 4 * g + e, M - 1) * rld);
  };
#ifdef JS2T_P192_PROF
  unsigned long long e_last_ = __builtin_readcyclecounter();
#endif
  if (has_res || has_gate) load_rg(0);
  P192_E(0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int mrow = mw + 16 * i + 4 * g;
    float ln_rs[4];
    if (lnf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ln_rs[e] = __shfl(my_rs, 16 * i + 4 * g + e);
    }
    if ((has_res || has_gate) && i < 2) load_rg(i + 1);
#ifdef JS2T_P192_PROF
    if (has_res || has_gate) asm volatile("s_nop 0" ::"v"(rg[i][3].w));  // waits for block i's rows
    P192_E(1 + 2 * i);
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = mrow + e;
      float v[8];
      if (lnf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(ln_rs[e], acc[i][j][e], bias_r[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = EPI < 0 ? acc[i][j][e] * alpha + bias_r[j] : (has_bias ? acc[i][j][e] + bias_r[j] : acc[i][j][e]);
      }
      if (relu) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      if (has_drop) {  // the decisions of dropout_keep4_key(drop_key, m, n/4 + h), taken straight from the hash halves
        const uint32_t rowkey = (share_keys ? row16_bcast(my_key, 4 * i + e) : hash32((uint32_t)m ^ drop_key)) + 2u * (uint32_t)(n >> 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t h = hash32w(rowkey + (uint32_t)q);
          v[2 * q] = (h & 0xffffu) >= thr ? v[2 * q] * keep_scale : 0.f;
          v[2 * q + 1] = (h >> 16) >= thr ? v[2 * q + 1] * keep_scale : 0.f;
        }
      }
      if (has_res) {
        float rr[8];
        unpack_bf16x8(rg[i][e], rr);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] += res_scale * rr[c];
      }
      if (has_gate) {
        float rr[8];
        unpack_bf16x8(rg[i][e], rr);
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = rr[c] > 0.f ? v[c] * gate_scale : 0.f;
      }
      if (m < M && (!OUT8 || d.C)) {
        uint4 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        pk.z = pack_bf16x2(v[4], v[5]);
        pk.w = pack_bf16x2(v[6], v[7]);
        *(uint4*)((uint16_t*)d.C + (int64_t)m * d.ldc + n) = pk;
      }
      if constexpr (OUT8) {
        if (d.c8 && m < M) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            q8_max = fmaxf(q8_max, fabsf(v[c]));
            v[c] = fminf(fmaxf(v[c] * q8_inv, -448.f), 448.f);
          }
          int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
          int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
          *(uint2*)(d.c8 + (int64_t)m * d.ldc8 + n) = make_uint2((uint32_t)lo, (uint32_t)hi);
        }
      }
    }
    P192_E(2 + 2 * i);
  }
  if constexpr (OUT8) {
    if (d.c8) {  // this tile's max |v| for the NEXT call's scale: posted only if it beats what is there (decayed by the consumer)
      q8_max = wave_max(q8_max);
      unsigned int* st = (unsigned int*)d.c8_state;
      if (lane == 0 && __float_as_uint(q8_max) > *(volatile unsigned int*)(st + 1)) atomicMax(st + 1, __float_as_uint(q8_max));
    }
  }
}

// one fragment pair -> accumulator: a bf16 MFMA over 32 k, or two e4m3 MFMAs over 32 k each (the fragment's two 8-byte halves)
template <bool FP8>
__device__ __forceinline__ f32x4_t p192_mma(const frag_i4& a, const frag_i4& b, f32x4_t c) {
  if constexpr (FP8) {
    typedef long l2_t __attribute__((ext_vector_type(2)));
    const l2_t al = __builtin_bit_cast(l2_t, a), bl = __builtin_bit_cast(l2_t, b);
    c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al[0], bl[0], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al[1], bl[1], c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
  }
}

// FP8: e4m3 operands (one byte per element).  The stage keeps its 128-byte rows, i.e. 128 k instead of 64; a lane's 16-byte
// fragment piece then holds the operands of TWO v_mfma_f32_16x16x32_fp8_fp8 (8 bytes each).  Which 8 of the stage's k a
// lane group supplies to which MFMA is a permutation of k that A and B share (both are read through the same chunk index
// 4 kk + g), so the sums are unchanged.  Same MFMA rate as bf16, half the L2 -> LDS bytes per flop - and that, not the
// matrix pipe, is what bounds this kernel.
template <int EPI, int NST = P_NST, bool FP8 = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_p192_kernel(js2t_gemm_desc d, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using elem_t = typename std::conditional<FP8, uint8_t, uint16_t>::type;
  constexpr int CSH = FP8 ? 4 : 3, KST = FP8 ? 128 : 64;  // log2(elements per 16-byte chunk), k per stage
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // wave-uniform: LDS-DMA destinations (M0) stay on the scalar unit
  const int M = d.M, N = d.N, K = d.K, nk = (K + KST - 1) / KST;
  const int ntiles = tiles_m * tiles_n, G = gridDim.x;
  const elem_t* Ab = (const elem_t*)d.A;
  const elem_t* Bb = (const elem_t*)d.B;
  const int64_t lda = d.lda, ldb = d.ldb;

  // ---- issue side: runs up to three stages ahead of the multiply side, possibly already in the next tile
  int iv = blockIdx.x, ik = 0, islot = 0;
  const elem_t* asrc[6];
  const elem_t* bsrc[4];
  const int r8 = lane >> 3, s8 = lane & 7;
  auto set_tile_src = [&](int v) {
    const int lid = xcd_remap(v, ntiles);
    const int m0 = (lid / tiles_n) * P_BM, n0 = (lid % tiles_n) * 128;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int row = (w * 6 + q) * 8 + r8;
      asrc[q] = Ab + (int64_t)min(m0 + row, M - 1) * lda + ((s8 ^ kc_key<false>(row)) << CSH);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // piece P = 4w + q of the B image: fragment index j = P >> 1, lanes r = 8 (P & 1) + r8
      const int P = w * 4 + q, r = (P & 1) * 8 + r8;
      bsrc[q] = Bb + (int64_t)min(n0 + 8 * r + (P >> 1), N - 1) * ldb + ((s8 ^ (r >> 1)) << CSH);
    }
  };
  // a stage is requested in ten 1 KB pieces per wave (six of A, four of B); the steady state spreads them between the
  // MFMAs of two k-halves, because four waves issuing forty DMA instructions back to back keep the MFMA pipe idle for
  // ~570 cycles (the texture-address path takes ~16 cycles per instruction).
  // Past the block's last stage the requests go on (re-reading k = 0 of the last tile into a free slot): the loop and
  // its vmcnt counts stay branch-free, at the price of three unused stages per block.
  int p_k0 = 0, p_klim = KST;
  unsigned char* p_st = smem;
  // k offset (inside a stage) of the lane's 8 elements: the same for every A piece, two values for the B pieces
  const int kofs_a = (s8 ^ (r8 & 7)) << CSH, kofs_b0 = (s8 ^ (r8 >> 1)) << CSH, kofs_b1 = (s8 ^ ((8 + r8) >> 1)) << CSH;
  const elem_t* zsrc = (const elem_t*)&g_zero16;
  auto issue_begin = [&]() {
    p_k0 = iv < ntiles ? ik * KST : 0;
    p_klim = K - p_k0;  // < KST only in the partial last stage of a tile (K % KST != 0): those k come from a zero constant
    p_st = smem + islot * P_STAGE;
  };
  auto issue_piece = [&](int q) {  // q is a compile-time constant at every call site
    if (q < 6) {
      const elem_t* sp = asrc[q < 6 ? q : 0] + p_k0;
      if (p_klim < KST) {
        if (kofs_a >= p_klim) sp = zsrc;
      }
      __builtin_amdgcn_global_load_lds((g_cvoid*)sp, (l_void*)(p_st + (w * 6 + q) * 1024), 16, 0, 0);
    } else {
      const elem_t* sp = bsrc[q >= 6 ? q - 6 : 0] + p_k0;
      if (p_klim < KST) {
        if ((((w * 4 + q - 6) & 1) ? kofs_b1 : kofs_b0) >= p_klim) sp = zsrc;
      }
      __builtin_amdgcn_global_load_lds((g_cvoid*)sp, (l_void*)(p_st + P_ATILE + (w * 4 + q - 6) * 1024), 16, 0, 0);
    }
  };
  auto issue_finish = [&]() {
    if (iv < ntiles && ++ik == nk) {
      ik = 0;
      iv += G;
      if (iv < ntiles) set_tile_src(iv);
    }
    islot = islot == NST - 1 ? 0 : islot + 1;
  };
  auto issue_next = [&]() {
    issue_begin();
#pragma unroll
    for (int q = 0; q < 10; ++q) issue_piece(q);
    issue_finish();
  };

  // ---- multiply side: wave w owns rows 48w .. 48w+47 and all 128 columns of the tile
  const int arow = w * 48 + (lane & 15);  // + 16 i
  const int br = lane & 15;
  // byte offsets of this lane's fragment pieces inside a stage, per k-half (the XOR keys do not depend on i / j)
  const int aoff0 = arow * 128 + ((g ^ kc_key<false>(arow)) << 4), aoff1 = arow * 128 + (((4 + g) ^ kc_key<false>(arow)) << 4);
  const int boff0 = P_ATILE + p192_b_granule(8 * br, g) * 16, boff1 = P_ATILE + p192_b_granule(8 * br, 4 + g) * 16;
  auto read_half = [&](const unsigned char* st, int ao, int bo, frag_i4 (&fm)[3], frag_i4 (&fn)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) fn[j] = *(const frag_i4*)(st + bo + j * 2048);
#pragma unroll
    for (int i = 0; i < 3; ++i) fm[i] = *(const frag_i4*)(st + ao + i * 2048);
  };
  // the same eleven reads in the order the next k-half consumes them, two per MFMA group q = 0..5 (one in the last):
  // issued between the MFMAs of the current half instead of as a burst in front of them
  auto read_part = [&](int q, const unsigned char* st, int ao, int bo, frag_i4 (&fm)[3], frag_i4 (&fn)[8]) {
    auto rn = [&](int j) { fn[j] = *(const frag_i4*)(st + bo + j * 2048); };
    auto rm = [&](int i) { fm[i] = *(const frag_i4*)(st + ao + i * 2048); };
    if (q == 0) { rn(0); rn(1); }
    if (q == 1) { rn(2); rn(3); }
    if (q == 2) { rm(0); rn(4); }
    if (q == 3) { rn(5); rn(6); }
    if (q == 4) { rn(7); rm(1); }
    if (q == 5) { rm(2); }
  };

  set_tile_src(iv);
  issue_next();
  issue_next();
  if (NST == 3) {
    issue_begin();
#pragma unroll
    for (int q = 0; q < 5; ++q) issue_piece(q);  // pieces 5..9 of stage 2 follow in the first k-half of stage 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PER + 5) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PER) : "memory");  // stage 0 landed, stage 1 in flight
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  frag_i4 fm0[3], fn0[8], fm1[3], fn1[8];
  f32x4_t acc[3][8];
  read_half(smem, aoff0, boff0, fm0, fn0);
  int cslot = 0;
#ifdef JS2T_P192_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  const bool any_drop = EPI < 0 ? d.dropout_p > 0.f : (EPI & PE_DROP) != 0;
  const uint32_t drop_key = any_drop ? dropout_key(d.rng_state, d.rng_stream) : 0u;
  bool stores_behind = false;
  if (FP8 && d.c8 && d.c8_scale_out && blockIdx.x == 0 && t == 0) {  // the scale the e4m3 output of THIS launch is written with
    const float S = d.c8_state[0];
    *d.c8_scale_out = (S > 0.f ? S : 1.f) * (d.c8_mul ? *d.c8_mul : 1.f);
  }
  for (int v = blockIdx.x; v < ntiles; v += G) {
    const int lid = xcd_remap(v, ntiles);
    const int tm0 = (lid / tiles_n) * P_BM, tn0 = (lid % tiles_n) * 128;
    float bias_r[8];
    p192_load_bias<EPI>(d, tn0 + 8 * (lane & 15), bias_r);
    float my_rs = 0.f;
    if (EPI >= 0 && (EPI & PE_LNF)) my_rs = lnf_row_rstd(d, tm0 + w * 48, lane, tn0 == 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nk; ++k) {  // this block's stages form one stream across its tiles
      // first k-half; pieces 5..9 of the stage requested at the previous barrier and the reads of the second half ride
      // between its MFMA groups
      const unsigned char* cst = smem + cslot * P_STAGE;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int i = q >> 1, j = (q & 1) * 4 + jj;
          acc[i][j] = p192_mma<FP8>(fm0[i], fn0[j], acc[i][j]);
        }
        if (NST == 3 && q < 5) issue_piece(5 + q);
        read_part(q, cst, aoff1, boff1, fm1, fn1);
      }
      if (NST == 3) issue_finish();
      const int nslot = cslot == NST - 1 ? 0 : cslot + 1;
      P192_T(0);
      // stage s + 1 must have landed: own part by the counted wait (stage s + 2 stays in flight), everybody's by the barrier.
      // Right after a tile whose rows were all stored (12 stores per lane, none skipped) those stores sit between the
      // two stages in the in-order count: letting them stay in flight too saves a write-acknowledge latency per tile.
      // Any other tile keeps the smaller count, which only waits longer.
      if (k == 0 && stores_behind) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST == 3 ? P_PER : 0) + 12) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST == 3 ? P_PER : 0) : "memory");
      }
      P192_T(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      P192_T(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      P192_T(3);
      issue_begin();  // stage s + 3 goes into the slot of stage s (every wave holds its second k-half in registers)
      P192_T(4);
      const unsigned char* nst = smem + nslot * P_STAGE;  // after the last stage: unused reads of a stale slot
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int i = q >> 1, j = (q & 1) * 4 + jj;
          acc[i][j] = p192_mma<FP8>(fm1[i], fn1[j], acc[i][j]);
        }
        if (NST == 3) {
          if (q < 5) issue_piece(q);
        } else if (q < 5) {  // the whole of stage s + 2, two requests per MFMA group
          issue_piece(2 * q);
          issue_piece(2 * q + 1);
        }
        read_part(q, nst, aoff0, boff0, fm0, fn0);
      }
      if (NST == 2) issue_finish();
      cslot = nslot;
      P192_T(5);
    }
    p192_store_tile<EPI, FP8>(d, acc, tm0 + w * 48, tn0, lane, bias_r, drop_key, my_rs);
    stores_behind = tm0 + P_BM <= M;
    P192_T(6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the unused tail requests still target this block's LDS
  if (FP8 && d.fp8_state && blockIdx.x == 0 && t == 0) {
    // delayed activation scale of the producer that quantised A (js2t_layernorm_fwd_fp8): this product sits between two of its
    // calls on the stream, so it does the hand-over - the maximum the last call collected becomes the next call's scale.
    // (This launch's own scale arrived in alpha_dev, a separate word.)
    // The collected maximum is not cleared but DECAYED (x 15/16): the next call's blocks then post their maximum only if it
    // comes within 6 % of the last one - a handful of atomics instead of one per block that finishes while the word still
    // reads 0 (measured: 12 us per LayerNorm launch); a shrinking activation range is followed at 6 % per call.
    const float am = d.fp8_state[1];
    if (am > 0.f) d.fp8_state[0] = am * (1.f / 448.f);
    d.fp8_state[1] = am * 0.9375f;
  }
#ifdef JS2T_P192_PROF
  if (blockIdx.x == 0 && t == 0)
    for (int i = 0; i < 8; ++i) g_p192_prof[i] = prof_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// The same tile and stage image with the work split between waves (js2t_gemm_p192_ring(4))
// ------------------------------------------------------------------------------------------------
// Two measurements behind this variant (tools/l2_feed_probe.hip, tools/mfma_rate_probe.hip, MI355X):
//  * a CU pulls a 40 KB stage out of its L2 in ~410-490 ns by LDS-DMA when nothing else is going on (fragments loaded
//    straight into registers are 2.7x slower: 16 rows x 64 B per instruction) - but a wave that meets a DMA request while
//    the texture path's queue is full stalls IN ORDER, and its MFMAs stall with it;
//  * ONE wave per SIMD issuing v_mfma_f32_16x16x32_bf16 with its fragment reads in between reaches 1.76 PFLOP/s chip-wide
//    (~20 cycles per MFMA), TWO waves per SIMD 2.1-2.2 (16 cycles, the pipe's rate at the ~2.1 GHz it runs at).
// So a block is 12 waves: waves 0-7 (two per SIMD) only read fragments, multiply and store - wave (wm, wn) owns rows
// 48 wm .. +47 and columns 64 wn .. +63 of the tile - and waves 8-11 only issue the requests, each the pieces wave w - 8
// of the kernel above issues.  Same 3-slot ring, one s_barrier per stage for all twelve waves:
//   loader   : [stage s+1 landed: counted vmcnt, stage s+2 stays in flight] -> barrier s -> request stage s+3 (slot of s)
//   consumer : MFMAs of k-half 0 of stage s + reads of its k-half 1 -> lgkmcnt(0) -> barrier s -> MFMAs of k-half 1 +
//              reads of k-half 0 of stage s+1
// B image: granule (j, r, c) as in p192_b_granule, but fragment j of lane r is column 64 (j >> 2) + 4 r + (j & 3): after
// its four MFMAs along N a lane holds 4 CONSECUTIVE columns and the 16 lanes of a row group store one 128-byte run.
// One block per CU (120 KB of LDS, 168 registers per wave).
// OUT8 (the e4m3 kernel only): d.c8 != NULL adds the second output of p192_store_tile - the result as e4m3 bytes with the delayed
// scale of d.c8_state, four bytes per row and lane; d.C may then be NULL.
template <int EPI, bool OUT8 = false>
__device__ __forceinline__ void p192s_store_tile(const js2t_gemm_desc& d, f32x4_t (&acc)[3][4], int mw, int n0, int lane,
                                                 const float (&bias_r)[4], uint32_t drop_key, float my_rs = 0.f) {
#ifdef JS2T_GEMM_NOEPI
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
  return;
#endif
  const int g = lane >> 4, r = lane & 15;
  const int M = d.M, n = n0 + 4 * r;
  if (n >= d.N) return;  // N is a multiple of 8 (hence of 4): a lane's column group lies inside or outside as a whole
  float q8_inv = 0.f, q8_max = 0.f;
  if constexpr (OUT8) {
    if (d.c8) {
      const float S = d.c8_state[0];
      q8_inv = S > 0.f ? 1.f / S : 0.f;
    }
  }
  const float alpha = EPI < 0 ? d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f) : 1.f;
  const bool has_bias = EPI < 0 ? d.bias != nullptr : (EPI & PE_BIAS) != 0;
  const bool relu = EPI < 0 ? d.act == JS2T_ACT_RELU : (EPI & PE_RELU) != 0;
  const bool has_res = EPI < 0 ? d.residual != nullptr : (EPI & PE_RES) != 0;
  const bool has_gate = EPI < 0 ? d.gate != nullptr : (EPI & PE_GATE) != 0;
  const bool has_drop = EPI < 0 ? d.dropout_p > 0.f : (EPI & PE_DROP) != 0;
  const float keep_scale = 1.f / (1.f - d.dropout_p), res_scale = d.res_scale, gate_scale = d.gate_scale;
  const uint32_t thr = (uint32_t)(d.dropout_p * 65536.0f);
  const uint16_t* rsrc = (const uint16_t*)(has_res ? d.residual : d.gate) + n;
  const int64_t rld = has_res ? d.ldr : d.ldg;
  // one row key per lane (the 16 lanes of a row group walk the same twelve rows), fetched by row16_bcast: a lane of this kernel owns
  // four outputs per row, so hashing every row itself was two of its ~twelve vector instructions per output (round 6)
  // (whole 64-column groups only: a lane outside N has returned above and cannot hand its key over)
  const bool share_keys = n0 + 64 <= d.N;
  const uint32_t my_key = (has_drop && share_keys) ? row16_own_key(mw, lane, drop_key) : 0u;
  uint2 rg[3][4];
  if (has_res || has_gate) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) rg[i][e] = *(const uint2*)(rsrc + (int64_t)min(mw + 16 * i + 4 * g + e, M - 1) * rld);
  }
  // the fold exists in the specialised instantiations only (launch_bf16_p192 refuses other combinations): the run-time
  // decided variant (EPI < 0) would otherwise carry both paths and spill
  constexpr bool lnf = EPI >= 0 && (EPI & PE_LNF) != 0, wstats = EPI >= 0 && (EPI & PE_STATS) != 0, wdot = EPI >= 0 && (EPI & PE_DOT) != 0;
  const int grp = n0 >> 6, ngrp = d.N >> 6;  // this wave's 64-column group of the row, of N / 64
  uint2 dg[3][4];  // PE_DOT: the rows of dot_src this lane's columns meet (js2t_gemm_desc.dot_partial)
  if (wdot) {
    const uint16_t* dsrc = (const uint16_t*)d.dot_src + n;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) dg[i][e] = *(const uint2*)(dsrc + (int64_t)min(mw + 16 * i + 4 * g + e, M - 1) * d.ld_dot);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float ln_rs[4];
    if (lnf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ln_rs[e] = __shfl(my_rs, 16 * i + 4 * g + e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = mw + 16 * i + 4 * g + e;
      float v[4];
      if (lnf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(ln_rs[e], acc[i][j][e], bias_r[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = EPI < 0 ? acc[i][j][e] * alpha + bias_r[j] : (has_bias ? acc[i][j][e] + bias_r[j] : acc[i][j][e]);
      }
      if (relu) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
      }
      if (has_drop) {  // the decisions of dropout_keep4_key(drop_key, m, n / 4)
        const uint32_t rowkey = (share_keys ? row16_bcast(my_key, 4 * i + e) : hash32((uint32_t)m ^ drop_key)) + 2u * (uint32_t)(n >> 2);
        const uint32_t h0 = hash32w(rowkey), h1 = hash32w(rowkey + 1u);
        v[0] = (h0 & 0xffffu) >= thr ? v[0] * keep_scale : 0.f;
        v[1] = (h0 >> 16) >= thr ? v[1] * keep_scale : 0.f;
        v[2] = (h1 & 0xffffu) >= thr ? v[2] * keep_scale : 0.f;
        v[3] = (h1 >> 16) >= thr ? v[3] * keep_scale : 0.f;
      }
      if (has_res || has_gate) {
        const uint2 q = rg[i][e];
        const float rr[4] = {__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                             __uint_as_float(q.y & 0xffff0000u)};
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = has_res ? v[c] + res_scale * rr[c] : (rr[c] > 0.f ? v[c] * gate_scale : 0.f);
      }
      if (wstats) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float q = bf16_round(v[c]);
          s1 += q;
          s2 = fmaf(q, q, s2);
        }
        s1 = row16_sum(s1), s2 = row16_sum(s2);
        if (r == 0 && m < M) *(float2*)(d.rs_partial + 2 * ((int64_t)m * ngrp + grp)) = make_float2(s1, s2);
      }
      if (wdot) {
        const uint2 q = dg[i][e];
        float sd = bf16_round(v[0]) * __uint_as_float(q.x << 16);
        sd = fmaf(bf16_round(v[1]), __uint_as_float(q.x & 0xffff0000u), sd);
        sd = fmaf(bf16_round(v[2]), __uint_as_float(q.y << 16), sd);
        sd = fmaf(bf16_round(v[3]), __uint_as_float(q.y & 0xffff0000u), sd);
        sd = row16_sum(sd);
        if (r == 0 && m < M) d.dot_partial[(int64_t)m * ngrp + grp] = sd;
      }
      if (m < M && (!OUT8 || d.C)) {
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)((uint16_t*)d.C + (int64_t)m * d.ldc + n) = pk;
      }
      if constexpr (OUT8) {
        if (d.c8 && m < M) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            q8_max = fmaxf(q8_max, fabsf(v[c]));
            v[c] = fminf(fmaxf(v[c] * q8_inv, -448.f), 448.f);
          }
          int q = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
          q = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], q, true);
          *(uint32_t*)(d.c8 + (int64_t)m * d.ldc8 + n) = (uint32_t)q;
        }
      }
    }
  }
  if constexpr (OUT8) {
    if (d.c8) {  // this tile's max |v| for the NEXT call's scale: posted only if it beats what is there (decayed by the consumer)
      q8_max = wave_max(q8_max);
      unsigned int* st = (unsigned int*)d.c8_state;
      if (lane == 0 && __float_as_uint(q8_max) > *(volatile unsigned int*)(st + 1)) atomicMax(st + 1, __float_as_uint(q8_max));
    }
  }
}

#ifdef JS2T_P192S_DBG
// [0] ticks of block 0's consumer wave 0 from first to last instruction, [1] of those spent between lgkmcnt(0) and the end
// of the barrier, [2] loader wave 8: ticks in vmcnt wait, [3] in barrier, [4] issuing, [5] stages
__device__ unsigned long long g_p192s_prof[8];
extern "C" int js2t_debug_p192s_prof(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_p192s_prof), 64); }
#endif
#ifndef JS2T_FP8_NO_SCALED
constexpr bool P192S_SCALED = true;  // e4m3 consumers on v_mfma_scale_f32_16x16x128_f8f6f4 (-DJS2T_FP8_NO_SCALED: the K = 32 instruction, for A/B)
#else
constexpr bool P192S_SCALED = false;
#endif
template <int EPI, bool FP8 = false>
__global__ __launch_bounds__(768, 1) void gemm_bf16_p192s_kernel(js2t_gemm_desc d, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using elem_t = typename std::conditional<FP8, uint8_t, uint16_t>::type;
  constexpr int CSH = FP8 ? 4 : 3, KST = FP8 ? 128 : 64, NST = PS_NST;
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4;
  const int w12 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = d.M, N = d.N, K = d.K, nk = (K + KST - 1) / KST;
  const int ntiles = tiles_m * tiles_n, G = gridDim.x;
  const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
  const int nstages = my_tiles * nk;  // every wave of the block passes 1 + nstages barriers

  if (w12 >= 8) {
    // ------------------------------------------------------------------------------------------ loader waves
    const int w = w12 - 8;
    const elem_t* Ab = (const elem_t*)d.A;
    const elem_t* Bb = (const elem_t*)d.B;
    const int64_t lda = d.lda, ldb = d.ldb;
    int iv = blockIdx.x, ik = 0, islot = 0;
    const elem_t* asrc[6];
    const elem_t* bsrc[4];
    const int r8 = lane >> 3, s8 = lane & 7;
    auto set_tile_src = [&](int v) {
      const int lid = xcd_remap(v, ntiles);
      const int m0 = (lid / tiles_n) * P_BM, n0 = (lid % tiles_n) * 128;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int row = (w * 6 + q) * 8 + r8;
        asrc[q] = Ab + (int64_t)min(m0 + row, M - 1) * lda + ((s8 ^ kc_key<false>(row)) << CSH);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // piece P = 4w + q of the B image: fragment index j = P >> 1, lanes r = 8 (P & 1) + r8
        const int P = w * 4 + q, r = (P & 1) * 8 + r8, jf = P >> 1;
        bsrc[q] = Bb + (int64_t)min(n0 + 64 * (jf >> 2) + 4 * r + (jf & 3), N - 1) * ldb + ((s8 ^ (r >> 1)) << CSH);
      }
    };
    const int kofs_a = (s8 ^ (r8 & 7)) << CSH, kofs_b0 = (s8 ^ (r8 >> 1)) << CSH, kofs_b1 = (s8 ^ ((8 + r8) >> 1)) << CSH;
    const elem_t* zsrc = (const elem_t*)&g_zero16;
    // past the block's last stage the requests go on (k = 0 of its last tile into a free slot): branch-free counts
    auto issue_next = [&]() {
      const int k0 = iv < ntiles ? ik * KST : 0, klim = K - k0;
      unsigned char* st = smem + islot * P_STAGE;
#if defined(JS2T_P192S_DBG) && (JS2T_P192S_DBG & 1)  // measurement only: no requests, the consumers multiply stale LDS
      return;
#endif
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const elem_t* sp = asrc[q] + k0;
        if (klim < KST && kofs_a >= klim) sp = zsrc;
        __builtin_amdgcn_global_load_lds((g_cvoid*)sp, (l_void*)(st + (w * 6 + q) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const elem_t* sp = bsrc[q] + k0;
        if (klim < KST && (((w * 4 + q) & 1) ? kofs_b1 : kofs_b0) >= klim) sp = zsrc;
        __builtin_amdgcn_global_load_lds((g_cvoid*)sp, (l_void*)(st + P_ATILE + (w * 4 + q) * 1024), 16, 0, 0);
      }
      if (iv < ntiles && ++ik == nk) {
        ik = 0;
        iv += G;
        if (iv < ntiles) set_tile_src(iv);
      }
      islot = islot == NST - 1 ? 0 : islot + 1;
    };
    set_tile_src(min(iv, ntiles - 1));
#pragma unroll
    for (int q = 0; q < NST; ++q) issue_next();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * P_PER) : "memory");  // stage 0 landed
    __builtin_amdgcn_s_barrier();
#ifdef JS2T_P192S_DBG
    unsigned long long pw = 0, pb = 0, pi = 0, c0 = __builtin_readcyclecounter(), c1;
#define P192S_L(acc) do { c1 = __builtin_readcyclecounter(); acc += c1 - c0; c0 = c1; } while (0)
#else
#define P192S_L(acc)
#endif
    for (int sidx = 0; sidx < nstages; ++sidx) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * P_PER) : "memory");  // stage sidx + 1 landed, the later ones in flight
      P192S_L(pw);
      __builtin_amdgcn_s_barrier();                                  // ... and every consumer holds stage sidx in registers
      P192S_L(pb);
      issue_next();                                                  // stage sidx + NST into the slot of stage sidx
      P192S_L(pi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the unused tail requests still target this block's LDS
#ifdef JS2T_P192S_DBG
    if (blockIdx.x == 0 && t == 512) g_p192s_prof[2] = pw, g_p192s_prof[3] = pb, g_p192s_prof[4] = pi, g_p192s_prof[5] = nstages;
#endif
    return;
  }

  // -------------------------------------------------------------------------------------------- consumer waves
  const int wm = w12 & 3, wn = w12 >> 2;
  const int arow = wm * 48 + (lane & 15);
  const int br = lane & 15;
  const int aoff0 = arow * 128 + ((g ^ kc_key<false>(arow)) << 4), aoff1 = arow * 128 + (((4 + g) ^ kc_key<false>(arow)) << 4);
  const int boff0 = P_ATILE + wn * 8192 + p192_b_granule(8 * br, g) * 16, boff1 = P_ATILE + wn * 8192 + p192_b_granule(8 * br, 4 + g) * 16;
  auto read_half = [&](const unsigned char* st, int ao, int bo, frag_i4 (&fm)[3], frag_i4 (&fn)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) fn[j] = *(const frag_i4*)(st + bo + j * 2048);
#pragma unroll
    for (int i = 0; i < 3; ++i) fm[i] = *(const frag_i4*)(st + ao + i * 2048);
  };
  // the seven reads of the next k-half ride behind the first two of the three MFMA groups: the third group (and the other
  // wave of the SIMD) covers the latency of the last read, so neither the lgkmcnt(0) in front of the barrier nor the wait
  // in front of the next half's first MFMA finds anything outstanding
  auto read_part = [&](int q, const unsigned char* st, int ao, int bo, frag_i4 (&fm)[3], frag_i4 (&fn)[4]) {
#if defined(JS2T_P192S_DBG) && (JS2T_P192S_DBG & 4)  // measurement only: MFMAs on stale fragments, no LDS reads
    return;
#endif
    auto rn = [&](int j) { fn[j] = *(const frag_i4*)(st + bo + j * 2048); };
    auto rm = [&](int i) { fm[i] = *(const frag_i4*)(st + ao + i * 2048); };
    if (q == 0) { rm(0); rn(0); rn(1); rn(2); }
    if (q == 1) { rn(3); rm(1); rm(2); }
  };
  const bool any_drop = EPI < 0 ? d.dropout_p > 0.f : (EPI & PE_DROP) != 0;
  const uint32_t drop_key = any_drop ? dropout_key(d.rng_state, d.rng_stream) : 0u;
  __builtin_amdgcn_s_barrier();  // stage 0 landed
  asm volatile("" ::: "memory");
  f32x4_t acc[3][4];
  if constexpr (FP8 && P192S_SCALED) {
    // e4m3 on the block-scaled instruction: a stage (128 k) is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per accumulator (unit
    // scales, E8M0 127) instead of four v_mfma_f32_16x16x32_fp8_fp8 - twice the issue rate (tools/fp8_mfma_probe.hip: 4.87
    // against 2.35 PFLOP/s chip-wide).  A lane's 32 operand bytes are its two 16-byte k-half fragments of the stage, for A
    // and B alike, i.e. the same permutation of k on both sides.  All of a stage's fragments have to be in registers before
    // its MFMAs start, so the slot is free from then on: lgkmcnt(0) -> barrier -> 12 MFMAs with the 14 reads of the NEXT
    // stage between them (two register sets, alternating).
    struct Frags { frag_i4 m[2][3], n[2][4]; };
    Frags fx, fy;
    auto read_stage = [&](const unsigned char* st, Frags& f) {
      read_half(st, aoff0, boff0, f.m[0], f.n[0]);
      read_half(st, aoff1, boff1, f.m[1], f.n[1]);
    };
    auto cat = [](const frag_i4& lo, const frag_i4& hi) {
      typedef int i8_t __attribute__((ext_vector_type(8)));
      return i8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto run_stage = [&](Frags& cur, Frags& nxt, const unsigned char* nst) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this stage's fragments are all here: its slot is free
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat(cur.m[0][i], cur.m[1][i]), cat(cur.n[0][j], cur.n[1][j]), acc[i][j], 0,
                                                                       0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        // the next stage's fragments behind the MFMA groups: B first (it is live for the whole stage), A last, into the registers
        // this stage's A fragments leave (two full sets + 48 accumulators do not fit the 168 registers three waves per SIMD have)
        __builtin_amdgcn_sched_barrier(0);
        if (i < 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) nxt.n[i][j] = *(const frag_i4*)(nst + (i == 0 ? boff0 : boff1) + j * 2048);
        } else {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            nxt.m[0][q] = *(const frag_i4*)(nst + aoff0 + q * 2048);
            nxt.m[1][q] = *(const frag_i4*)(nst + aoff1 + q * 2048);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (d.c8 && d.c8_scale_out && blockIdx.x == 0 && t == 0) {  // the scale the e4m3 output of THIS launch is written with
      const float S = d.c8_state[0];
      *d.c8_scale_out = (S > 0.f ? S : 1.f) * (d.c8_mul ? *d.c8_mul : 1.f);
    }
    read_stage(smem, fx);
    // the block's stages as ONE stream, two per iteration: which register set is the current one is then known statically
    // (chosen by a run-time parity both sets stay live across the loop: 112 + 48 registers and spills)
    int cslot = 0, v = blockIdx.x, k = 0, tm0 = 0, tn0 = 0;
    float bias_r[4] = {0.f, 0.f, 0.f, 0.f}, my_rs = 0.f;
    auto begin_tile = [&]() {
      const int lid = xcd_remap(v, ntiles);
      tm0 = (lid / tiles_n) * P_BM, tn0 = (lid % tiles_n) * 128 + 64 * wn;
      const int n = tn0 + 4 * br;
      const bool has_bias = (EPI < 0 ? d.bias != nullptr : (EPI & PE_BIAS) != 0) && n < N;
#pragma unroll
      for (int c = 0; c < 4; ++c) bias_r[c] = has_bias ? d.bias[n + c] : 0.f;
      if (EPI >= 0 && (EPI & PE_LNF)) my_rs = lnf_row_rstd(d, tm0 + wm * 48, lane, tn0 == 0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    auto one = [&](Frags& cur, Frags& nxt) {
      const int nslot = cslot == NST - 1 ? 0 : cslot + 1;
      run_stage(cur, nxt, smem + nslot * P_STAGE);  // after the block's last stage: unused reads of a stale slot
      cslot = nslot;
      if (++k == nk) {
        p192s_store_tile<EPI, true>(d, acc, tm0 + wm * 48, tn0, lane, bias_r, drop_key, my_rs);
        k = 0, v += G;
        if (v < ntiles) begin_tile();
      }
    };
    if (v < ntiles) begin_tile();
    int sidx = 0;
    for (; sidx + 1 < nstages; sidx += 2) {
      one(fx, fy);
      one(fy, fx);
    }
    if (sidx < nstages) one(fx, fy);
    if (d.fp8_state && blockIdx.x == 0 && t == 0) {  // delayed activation scale of A's producer: see gemm_bf16_p192_kernel
      const float am = d.fp8_state[1];
      if (am > 0.f) d.fp8_state[0] = am * (1.f / 448.f);
      d.fp8_state[1] = am * 0.9375f;
    }
    return;
  }
  frag_i4 fm0[3], fn0[4], fm1[3], fn1[4];
  read_half(smem, aoff0, boff0, fm0, fn0);
  int cslot = 0;
#ifdef JS2T_P192S_DBG
  unsigned long long cons_bar = 0, cons_epi = 0;
  const unsigned long long cons_t0 = __builtin_readcyclecounter();
#endif
  for (int v = blockIdx.x; v < ntiles; v += G) {
    const int lid = xcd_remap(v, ntiles);
    const int tm0 = (lid / tiles_n) * P_BM, tn0 = (lid % tiles_n) * 128 + 64 * wn;
    float bias_r[4];
    {
      const int n = tn0 + 4 * br;
      const bool has_bias = (EPI < 0 ? d.bias != nullptr : (EPI & PE_BIAS) != 0) && n < N;
#pragma unroll
      for (int c = 0; c < 4; ++c) bias_r[c] = has_bias ? d.bias[n + c] : 0.f;
    }
    float my_rs = 0.f;
    if (EPI >= 0 && (EPI & PE_LNF)) my_rs = lnf_row_rstd(d, tm0 + wm * 48, lane, tn0 == 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nk; ++k) {
      const unsigned char* cst = smem + cslot * P_STAGE;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if defined(JS2T_P192S_DBG) && (JS2T_P192S_DBG & 2)  // measurement only: reads without MFMAs
          asm volatile("" ::"v"(fm0[i]), "v"(fn0[j]));
#else
          acc[i][j] = p192_mma<FP8>(fm0[i], fn0[j], acc[i][j]);
#endif
        }
        // scheduling fences: the reads stay behind their MFMA group (the scheduler would sink them into one burst, or hoist
        // them above the group - and then the wait for this group's operands also waits for them)
        __builtin_amdgcn_sched_barrier(0);
        read_part(i, cst, aoff1, boff1, fm1, fn1);
        __builtin_amdgcn_sched_barrier(0);
      }
      const int nslot = cslot == NST - 1 ? 0 : cslot + 1;
#ifdef JS2T_P192S_DBG
      const unsigned long long cb0 = __builtin_readcyclecounter();
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the last read of this slot has returned
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef JS2T_P192S_DBG
      cons_bar += __builtin_readcyclecounter() - cb0;
#endif
      const unsigned char* nst = smem + nslot * P_STAGE;  // after the block's last stage: unused reads of a stale slot
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if defined(JS2T_P192S_DBG) && (JS2T_P192S_DBG & 2)
          asm volatile("" ::"v"(fm1[i]), "v"(fn1[j]));
#else
          acc[i][j] = p192_mma<FP8>(fm1[i], fn1[j], acc[i][j]);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        read_part(i, nst, aoff0, boff0, fm0, fn0);
        __builtin_amdgcn_sched_barrier(0);
      }
      cslot = nslot;
    }
#ifdef JS2T_P192S_DBG
    const unsigned long long ce0 = __builtin_readcyclecounter();
#endif
    p192s_store_tile<EPI>(d, acc, tm0 + wm * 48, tn0, lane, bias_r, drop_key, my_rs);
#ifdef JS2T_P192S_DBG
    cons_epi += __builtin_readcyclecounter() - ce0;
#endif
  }
  if (FP8 && d.fp8_state && blockIdx.x == 0 && t == 0) {  // delayed activation scale of A's producer: see gemm_bf16_p192_kernel
    const float am = d.fp8_state[1];
    if (am > 0.f) d.fp8_state[0] = am * (1.f / 448.f);
    d.fp8_state[1] = am * 0.9375f;
  }
#ifdef JS2T_P192S_DBG
  if (blockIdx.x == 0 && t == 0)
    g_p192s_prof[0] = __builtin_readcyclecounter() - cons_t0, g_p192s_prof[1] = cons_bar, g_p192s_prof[6] = cons_epi;
#endif
}

// ------------------------------------------------------------------------------------------------
// The same persistent 192x128 pipeline for reduction-major operands: grouped weight gradients
// C_g[M,N] (f32) = alpha * A_g[K,M]^T B_g[K,N] + beta * C_g, optional row sums of A_g^T (bias gradients)
// ------------------------------------------------------------------------------------------------
// Stage = A^T image [64 k][192 m] split in a 128-wide part (256-byte rows, the XOR key of the two-stage kernel) and a
// 64-wide part (128-byte rows, two k rows per 256-byte bank window, key from k bits 1 and 3), plus the B^T image
// [64 k][128 n]; every fragment is two ds_read_b64_tr_b16 (4 k each).  Natural fragment columns: lane (g, r) ends
// up with C[16i + 4g + e][16j + r] - 16 lanes store 64 contiguous bytes of f32; the epilogue is a read-modify-write of
// single dwords, which only a K of thousands amortises (this form is used for K = tokens).  A partial last stage takes
// its k >= K rows from a 16-byte zero constant.  Blocks walk the tiles of all group members in one XCD-aware order.
constexpr int PT_A0 = 0, PT_A1 = 16384, PT_B = 24576;

template <bool RS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_p192t_kernel(js2t_gemm_desc d, GemmGroup grp, int tiles_m, int tiles_n, int count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, r = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = d.M, N = d.N, K = d.K, nk = (K + 63) >> 6;
  const int per_member = tiles_m * tiles_n, ntiles = per_member * count, G = gridDim.x;
  const int64_t lda = d.lda, ldb = d.ldb;
  const uint16_t* zsrc = (const uint16_t*)&g_zero16;

  // ---- issue side
  int iv = blockIdx.x, ik = 0, islot = 0;
  const uint16_t* src[10];   // 4 pieces of A0, 2 of A1, 4 of B: address of the lane's 16 bytes at k = 0
  // k row (inside a stage) the lane's granule of piece q belongs to
  auto krow = [&](int q) { return q < 4 ? (w * 4 + q) * 4 + (lane >> 4) : q < 6 ? (w * 2 + q - 4) * 8 + (lane >> 3) : (w * 4 + q - 6) * 4 + (lane >> 4); };
  auto set_tile_src = [&](int v) {
    const int lid = xcd_remap(v, ntiles);
    const int mem = lid / per_member, rem = lid - mem * per_member;
    const int m0 = (rem / tiles_n) * P_BM, n0 = (rem % tiles_n) * 128;
    const uint16_t* Ab = (const uint16_t*)grp.A[mem];
    const uint16_t* Bb = (const uint16_t*)grp.B[mem];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kr = krow(q);
      int c = m0 + (((lane & 15) ^ (tr_g(kr) << 1)) << 3);
      if (c >= M) c = m0;  // columns that are never stored
      src[q] = Ab + (int64_t)kr * lda + c;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kr = krow(4 + q);
      const int key2 = ((kr >> 1) & 1) | (((kr >> 3) & 1) << 1);
      int c = m0 + 128 + (((lane & 7) ^ (key2 << 1)) << 3);
      if (c >= M) c = m0;
      src[4 + q] = Ab + (int64_t)kr * lda + c;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kr = krow(6 + q);
      src[6 + q] = Bb + (int64_t)kr * ldb + n0 + (((lane & 15) ^ (tr_g(kr) << 1)) << 3);
    }
  };
  int64_t p_ka = 0, p_kb = 0;
  int p_klim = 64;  // valid k rows of the stage being requested
  unsigned char* p_st = smem;
  auto issue_begin = [&]() {
    const int k0 = iv < ntiles ? ik << 6 : 0;
    p_ka = (int64_t)k0 * lda;
    p_kb = (int64_t)k0 * ldb;
    p_klim = K - k0;
    p_st = smem + islot * P_STAGE;
  };
  auto issue_piece = [&](int q) {  // q is a compile-time constant at every call site
    const uint16_t* sp = src[q] + (q < 6 ? p_ka : p_kb);
    if (p_klim < 64) {  // partial last stage (wave-uniform test): rows k >= K come from the zero constant
      if (krow(q) >= p_klim) sp = zsrc;
    }
    unsigned char* dst = q < 4 ? p_st + PT_A0 + (w * 4 + q) * 1024
                       : q < 6 ? p_st + PT_A1 + (w * 2 + q - 4) * 1024 : p_st + PT_B + (w * 4 + q - 6) * 1024;
    lds_dma16(sp, dst);
  };
  auto issue_finish = [&]() {
    if (iv < ntiles && ++ik == nk) {
      ik = 0;
      iv += G;
      if (iv < ntiles) set_tile_src(iv);
    }
    islot = islot == P_NST - 1 ? 0 : islot + 1;
  };
  auto issue_next = [&]() {
    issue_begin();
#pragma unroll
    for (int q = 0; q < 10; ++q) issue_piece(q);
    issue_finish();
  };

  // ---- multiply side: wave w owns the 16-row blocks w, w + 4 (128-wide part) and w + 8 (64-wide part) of the tile and
  // all 128 columns, so that the image a fragment comes from is known at compile time.
  // Per lane the XOR keys are constants: k = 32 kk + 8 g + q4 (+4 for the second read) has k & 3 = q4, (k >> 3) & 1 = g & 1,
  // (k >> 1) & 1 = q4 >> 1.  offB / offA = byte offset inside a stage of the lane's first 8-byte piece at kk = 0; the
  // other three pieces of a fragment pair are at immediate offsets (+4 k rows, +32 k rows).
  const int q4 = r >> 2, p4 = r & 3;
  const int key3 = q4 | ((g & 1) << 2), key2 = (q4 >> 1) | ((g & 1) << 1);
  int offB[8], offA[3];
#pragma unroll
  for (int j = 0; j < 8; ++j) offB[j] = PT_B + (8 * g + q4) * 256 + (((16 * j) * 2 + 8 * p4) ^ (key3 << 5));
#pragma unroll
  for (int i = 0; i < 2; ++i) offA[i] = PT_A0 + (8 * g + q4) * 256 + (((16 * (w + 4 * i)) * 2 + 8 * p4) ^ (key3 << 5));
  offA[2] = PT_A1 + (8 * g + q4) * 128 + (((16 * w) * 2 + 8 * p4) ^ (key2 << 5));
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  auto tr_pair = [&](const unsigned char* p0, const unsigned char* p1) -> frag_i4 {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p1);
    const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(frag_i4, v);
  };
  auto read_part = [&](int q, const unsigned char* st, int kk, frag_i4 (&fm)[3], frag_i4 (&fn)[8]) {
    auto rn = [&](int j) { fn[j] = tr_pair(st + offB[j] + kk * 8192, st + offB[j] + kk * 8192 + 1024); };
    auto rm = [&](int i) {
      fm[i] = i < 2 ? tr_pair(st + offA[i] + kk * 8192, st + offA[i] + kk * 8192 + 1024)
                    : tr_pair(st + offA[2] + kk * 4096, st + offA[2] + kk * 4096 + 512);
    };
    if (q == 0) { rn(0); rn(1); }
    if (q == 1) { rn(2); rn(3); }
    if (q == 2) { rm(0); rn(4); }
    if (q == 3) { rn(5); rn(6); }
    if (q == 4) { rn(7); rm(1); }
    if (q == 5) { rm(2); }
  };

  set_tile_src(iv);
  issue_next();
  issue_next();
  issue_begin();
#pragma unroll
  for (int q = 0; q < 5; ++q) issue_piece(q);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PER + 5) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  frag_i4 fm0[3], fn0[8], fm1[3], fn1[8];
  f32x4_t acc[3][8], racc[3];
#pragma unroll
  for (int q = 0; q < 6; ++q) read_part(q, smem, 0, fm0, fn0);
  const frag_i4 ones = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
  int cslot = 0;
#ifdef JS2T_P192_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f), beta = d.beta;
  for (int v = blockIdx.x; v < ntiles; v += G) {
    const int lid = xcd_remap(v, ntiles);
    const int mem = lid / per_member, rem = lid - mem * per_member;
    const int tm0 = (rem / tiles_n) * P_BM, tn0 = (rem % tiles_n) * 128;
    const bool do_rs = RS && tn0 == 0 && grp.rowsum[mem] != nullptr;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      racc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    for (int k = 0; k < nk; ++k) {
      const unsigned char* cst = smem + cslot * P_STAGE;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int i = q >> 1, j = (q & 1) * 4 + jj;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(fm0[i]), as_bf16x8(fn0[j]), acc[i][j], 0, 0, 0);
        }
        if (q < 5) issue_piece(5 + q);
        read_part(q, cst, 1, fm1, fn1);
      }
      if (RS && do_rs) {
#pragma unroll
        for (int i = 0; i < 3; ++i) racc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(fm0[i]), as_bf16x8(ones), racc[i], 0, 0, 0);
      }
      issue_finish();
      const int nslot = cslot == P_NST - 1 ? 0 : cslot + 1;
      P192_T(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P_PER) : "memory");
      P192_T(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      P192_T(2);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      P192_T(3);
      issue_begin();
      P192_T(4);
      const unsigned char* nst = smem + nslot * P_STAGE;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int i = q >> 1, j = (q & 1) * 4 + jj;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(fm1[i]), as_bf16x8(fn1[j]), acc[i][j], 0, 0, 0);
        }
        if (q < 5) issue_piece(q);
        read_part(q, nst, 0, fm0, fn0);
      }
      if (RS && do_rs) {
#pragma unroll
        for (int i = 0; i < 3; ++i) racc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(fm1[i]), as_bf16x8(ones), racc[i], 0, 0, 0);
      }
      cslot = nslot;
      P192_T(5);
    }
    // ---- epilogue: C[m][n] = alpha * acc + beta * C, one dword per lane and (i, e, j); rows beyond M are skipped
    float* Cb = (float*)grp.C[mem];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = tm0 + 16 * (w + 4 * i) + 4 * g + e;
        if (m < M) {
          float* crow = Cb + (int64_t)m * d.ldc + tn0 + r;
          float old[8];
          if (beta != 0.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) old[j] = crow[16 * j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) crow[16 * j] = beta != 0.f ? acc[i][j][e] * alpha + beta * old[j] : acc[i][j][e] * alpha;
        }
      }
    }
    if (RS && do_rs && r == 0) {
      typedef __attribute__((address_space(1))) float gfloat;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = tm0 + 16 * (w + 4 * i) + 4 * g + e;
          if (m < M) __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)grp.rowsum[mem] + m, racc[i][e]);
        }
    }
  }
  P192_T(6);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef JS2T_P192_PROF
  if (blockIdx.x == 0 && t == 0)
    for (int i = 0; i < 8; ++i) g_p192_prof[i] = prof_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// 256x128x64 tile, 8 multiplying waves of 64x64 + 4 requesting waves, three-slot DMA ring: the big grouped weight gradients
// C_g[M,N] (f32) = alpha * A_g[K,M]^T B_g[K,N] + beta * C_g, optional row sums of A_g^T, optional tile sums of squares
// ------------------------------------------------------------------------------------------------
// The two-stage 128x128 kernel pulls 32 KB from L2 per 2.1 MFLOP and every wave requests its share of a stage itself.  This tile
// moves 48 KB per 4.2 MFLOP (0.75 x the bytes per flop), keeps TWO stages in flight behind the one being multiplied, and
// separates the roles: a first version whose eight waves requested their own pieces (six each, in the second k-half) ran at
// 2150 clocks per stage against 1024 of MFMA work - a global_load_lds occupies its wave for ~18 clocks per KB while the CU's
// one address path serialises the 48 requests of a stage, so the wave served first stood 650 clocks at the barrier waiting for
// the one served last, and nobody multiplied meanwhile (profiles/r05_wg256_v1_selfloading_*.txt: x1.00 - 1.09 over 128x128).
//  * stage = two [64 k][128 m] images of A^T (columns m0.. and m0 + 128..) + one [64 k][128 n] image of B^T, the 256-byte-row
//    layout of the two-stage kernel (XOR key tr_g(k) on the 32-byte column groups, two ds_read_b64_tr_b16 per fragment);
//    requesting wave l of 4 sends the four 1 KB pieces 4l .. 4l + 3 of each image (12 requests per stage); rows k >= K of a
//    partial last stage come from a 16-byte zero constant;
//  * one raw s_barrier per stage for all twelve waves, in the MIDDLE of the multiplying waves' stage (the scheme of
//    gemm_bf16_p192t_kernel): by then they hold the second k-half of stage s in registers, and a requesting wave arrives once
//    its counted vmcnt has retired its pieces of stage s + 1; behind the barrier slot s % 3 takes stage s + 3 while the second
//    half of s and the first half of s + 1 are multiplied;
//  * fragments are double-buffered by k-half, the order "four products, four transposing reads" is pinned by scheduling barriers
//    (left alone the compiler issues sixteen products, then sixteen reads, and waits); requests past the last stage re-read
//    stage 0 into a dead slot so that the counted waits need no branches;
//  * multiplying wave (wm, wn) of 4 x 2 owns C rows 64 wm .. + 63, columns 64 wn .. + 63: the accumulator layout, the MFMA
//    operand order and the k order are those of the 128x128 kernel, so the products are bit-identical to it; a tile is >= 100 us
//    of work, its epilogue (16-byte f32 pieces straight from the registers, read-modify-write for beta) does not matter.
// M % 256 == 0, N % 128 == 0 (weight shapes); one tile per block, blocks of a member neighbours in the XCD-aware order.
constexpr int WG_IMG = 16384, WG_STAGE = 3 * WG_IMG, WG_NST = 3, WG_LDS = WG_NST * WG_STAGE, WG_PER = 12;

// SPLITK: the launch is (member, K slice, tile); a block multiplies stages kt0 .. kt0 + nk of its slice and ADDS alpha * partial
// into a zero-filled C with row-contiguous f32 atomics (tile staged through the idle ring), like the 128x128 kernel's slices.
template <bool RS, bool SPLITK>
__global__ __launch_bounds__(768, 1) void gemm_bf16_wg256_kernel(js2t_gemm_desc d, GemmGroup grp, int tiles_m, int tiles_n, int split_k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int tiles = tiles_m * tiles_n;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int vm = lid / tiles, tile = lid - vm * tiles;  // vm = member * split_k + slice
  const int mem = SPLITK ? vm / split_k : vm, slice = SPLITK ? vm - mem * split_k : 0;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 128;
  const int K = d.K, nk_all = (K + 63) >> 6;
  const int per = SPLITK ? (nk_all + split_k - 1) / split_k : nk_all;
  const int kt0 = slice * per;
  const int nk = min(per, nk_all - kt0);  // >= 1: the launcher keeps split_k <= nk_all / 4
  float ss = 0.f;

  if (w >= 8) {
    // ---- requesting waves
    const int l = w - 8;
    const int64_t lda = d.lda, ldb = d.ldb;
    const uint16_t* zsrc = (const uint16_t*)&g_zero16;
    const uint16_t* src[12];  // this lane's 16 bytes of pieces 4l + q of A0 (0-3), A1 (4-7), B (8-11) at k = 0
    int krow[4];
    {
      const uint16_t* Ab = (const uint16_t*)grp.A[mem];
      const uint16_t* Bb = (const uint16_t*)grp.B[mem];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = (l * 4 + q) * 64 + lane, kr = p >> 4, col = (((p & 15) ^ (tr_g(kr) << 1)) << 3);
        krow[q] = kr;
        src[q] = Ab + (int64_t)kr * lda + m0 + col;
        src[4 + q] = Ab + (int64_t)kr * lda + m0 + 128 + col;
        src[8 + q] = Bb + (int64_t)kr * ldb + n0 + col;
      }
    }
    auto issue_stage = [&](int st, int slot) {  // st, slot wave-uniform
      const int k0 = st < nk ? (kt0 + st) << 6 : 0;  // past the end: the matrix's stage 0 again, into a slot nobody reads
      const int64_t ka = (int64_t)k0 * lda, kb = (int64_t)k0 * ldb;
      const int klim = K - k0;
      unsigned char* stp = smem + slot * WG_STAGE + l * 4096;
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const uint16_t* sp = src[q] + (q < 8 ? ka : kb);
        if (klim < 64) {  // partial last stage (wave-uniform test)
          if (krow[q & 3] >= klim) sp = zsrc;
        }
        lds_dma16(sp, stp + (q >> 2) * WG_IMG + (q & 3) * 1024);
      }
    };
    issue_stage(0, 0);
    issue_stage(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_PER) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_stage(2, 2);
    int cslot = 0;
#ifdef JS2T_P192_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nk; ++s) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_PER) : "memory");  // own pieces of stage s + 1 (stage s + 2 stays in flight)
      P192_T(4);
      __builtin_amdgcn_s_barrier();                                  // the multiplying waves are through with slot s % 3
      asm volatile("" ::: "memory");
      P192_T(5);
      issue_stage(s + 3, cslot);
      P192_T(6);
      cslot = cslot == WG_NST - 1 ? 0 : cslot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests past the end
#ifdef JS2T_P192_PROF
    if (blockIdx.x == 0 && t == 512)
      for (int i = 4; i < 8; ++i) g_p192_prof[i] = prof_[i];
#endif
  } else {
    // ---- multiplying waves
    const int wm = w >> 1, wn = w & 1;
    const int a_img = (wm >> 1) * WG_IMG, a_sub = (wm & 1) * 64, b_sub = wn * 64;
    bf16x8_t fm0[4], fn0[4], fm1[4], fn1[4];
    f32x4_t acc[4][4], racc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      racc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_rs = RS && n0 == 0 && wn == 0 && grp.rowsum[mem] != nullptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8_ones_t;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, s16x8_ones_t{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});

    __builtin_amdgcn_s_barrier();  // stage 0 has landed
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fm0[i] = frag_load2<true>(smem + a_img, a_sub + 16 * i, 0, lane);
      fn0[i] = frag_load2<true>(smem + 2 * WG_IMG, b_sub + 16 * i, 0, lane);
    }
    int cslot = 0;
#ifdef JS2T_P192_PROF
    unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    for (int s = 0; s < nk; ++s) {
      const unsigned char* cst = smem + cslot * WG_STAGE;
      // first k-half of stage s; its second half's fragments are read underneath, B's first: the next half's first products want
      // all of B and one block of A
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn0[j], fm0[i], acc[i][j], 0, 0, 0);
        if (i < 2) {
          fn1[2 * i] = frag_load2<true>(cst + 2 * WG_IMG, b_sub + 32 * i, 1, lane);
          fn1[2 * i + 1] = frag_load2<true>(cst + 2 * WG_IMG, b_sub + 32 * i + 16, 1, lane);
        } else {
          fm1[2 * i - 4] = frag_load2<true>(cst + a_img, a_sub + 32 * (i - 2), 1, lane);
          fm1[2 * i - 3] = frag_load2<true>(cst + a_img, a_sub + 32 * (i - 2) + 16, 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (RS && do_rs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) racc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm0[i], racc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      P192_T(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // own reads of slot s % 3
      P192_T(1);
      __builtin_amdgcn_s_barrier();                       // stage s + 1 has landed; slot s % 3 goes to the requesting waves
      asm volatile("" ::: "memory");
      P192_T(2);
      __builtin_amdgcn_sched_barrier(0);
      const int nslot = cslot == WG_NST - 1 ? 0 : cslot + 1;
      const unsigned char* nst = smem + nslot * WG_STAGE;
      // second k-half of stage s; the first half of stage s + 1 is read underneath
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn1[j], fm1[i], acc[i][j], 0, 0, 0);
        if (i < 2) {
          fn0[2 * i] = frag_load2<true>(nst + 2 * WG_IMG, b_sub + 32 * i, 0, lane);
          fn0[2 * i + 1] = frag_load2<true>(nst + 2 * WG_IMG, b_sub + 32 * i + 16, 0, lane);
        } else {
          fm0[2 * i - 4] = frag_load2<true>(nst + a_img, a_sub + 32 * (i - 2), 0, lane);
          fm0[2 * i - 3] = frag_load2<true>(nst + a_img, a_sub + 32 * (i - 2) + 16, 0, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (RS && do_rs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) racc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fm1[i], racc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      P192_T(3);
      cslot = nslot;
    }
#ifdef JS2T_P192_PROF
    if (blockIdx.x == 0 && t == 0)
      for (int i = 0; i < 4; ++i) g_p192_prof[i] = prof_[i];
#endif

    if (RS && do_rs && (lane >> 4) == 0) {  // every output row of the ones-product holds the same sums: take row 0
      typedef __attribute__((address_space(1))) float gfloat;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_atomic_fadd_f32((gfloat*)grp.rowsum[mem] + m0 + wm * 64 + 16 * i + lane, racc[i][0]);
    }
    if constexpr (SPLITK) {
      // the partial tile goes through the ring (idle by now) as [256 rows][32 float4 slots, slot ^= row & 7]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();  // (with the requesting waves, which have drained their requests: nothing lands in the ring any more)
      float* Cs = (float*)smem;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ml = wm * 64 + 16 * i + (lane & 15), n4 = wn * 16 + 4 * j + (lane >> 4);
          *(f32x4_t*)(Cs + ml * 128 + ((n4 ^ (ml & 7)) << 2)) = acc[i][j];
        }
    } else {
    // ---- epilogue: lane (g = lane >> 4, r = lane & 15) holds C[16 i + r][16 j + 4 g .. + 3] of its wave's 64 x 64
    const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f), beta = d.beta;
    float* Cb = (float*)grp.C[mem] + (int64_t)(m0 + wm * 64 + (lane & 15)) * d.ldc + n0 + wn * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* crow = Cb + (int64_t)(16 * i) * d.ldc;
      float4 old[4];
      if (beta != 0.f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) old[j] = *(const float4*)(crow + 16 * j);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 v = make_float4(acc[i][j][0] * alpha, acc[i][j][1] * alpha, acc[i][j][2] * alpha, acc[i][j][3] * alpha);
        if (beta != 0.f) v.x += beta * old[j].x, v.y += beta * old[j].y, v.z += beta * old[j].z, v.w += beta * old[j].w;
        *(float4*)(crow + 16 * j) = v;
        ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
      }
    }
    }
  }
  if constexpr (SPLITK) {
    // f32 atomics as whole 256-byte row segments, one row half per wave-instruction, all twelve waves (scattered 4-byte atomics
    // run an order of magnitude below the contiguous rate)
    if (w >= 8) __syncthreads();  // the requesting waves' side of the barrier in front of the staging writes
    __syncthreads();
    const float alpha = d.alpha * (d.alpha_dev ? *d.alpha_dev : 1.f);
    const float* Cs = (const float*)smem;
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat* cg = (gfloat*)grp.C[mem] + (int64_t)m0 * d.ldc + n0;
    for (int idx = w; idx < 512; idx += 12) {
      const int ml = idx >> 1, nl = (idx & 1) * 64 + lane;
      const float v = Cs[ml * 128 + ((((nl >> 2) ^ (ml & 7)) << 2) | (nl & 3))];
      __builtin_amdgcn_global_atomic_fadd_f32(cg + (int64_t)ml * d.ldc + nl, v * alpha);
    }
    return;
  }
  if (d.sumsq_partial) {  // two entries per block: the launch is accounted in 128 x 128 tiles (js2t_gemm_grouped_blocks)
    ss = block_sum(ss, (float*)smem);  // (starts with a barrier: every wave's requests have landed, every fragment has been read)
    if (t == 0) d.sumsq_partial[2 * blockIdx.x] = ss, d.sumsq_partial[2 * blockIdx.x + 1] = 0.f;
  }
}

// 0 = never, 1 = whenever the product qualifies, -1 = when it qualifies and fills the chip (default)
#define g_p192_mode js2t_ctx_value(JS2T_CTX_GEMM_P192_MODE)
inline bool p192_eligible(const js2t_gemm_desc& d) {
  if (g_p192_mode == 0) return false;
  if (d.trans_a || d.trans_b || d.conv || d.split_k > 1 || d.batch != 1 || d.dtype_c != JS2T_BF16) return false;
  if ((d.N & 7) || d.N < 128 || (d.K & 7) || d.K < 192 || d.M < 1) return false;
  if (d.preact || d.beta != 0.f || !(d.act == JS2T_ACT_NONE || d.act == JS2T_ACT_RELU) || (d.residual && d.gate)) return false;
  if ((((uintptr_t)d.C) & 15) || (d.ldc & 7)) return false;
  if (d.residual && ((d.ldr & 7) || (((uintptr_t)d.residual) & 15))) return false;
  if (d.gate && ((d.ldg & 7) || (((uintptr_t)d.gate) & 15))) return false;
  if (g_p192_mode < 0 && (int64_t)cdiv(d.M, P_BM) * (d.N >> 7) < 200) return false;
  return true;
}
// variant: 3 / 2 / 4 forced, -1 (default) by shape (measured on MI355X, tools/p192_ring_ab.py):
//  * at least one and a half tiles per CU -> two blocks per CU on two-slot rings (QKV x1.15, FFN1 x1.23, ReLU-gated input
//    gradient x1.23, CTC projection x1.11 over the single block with a three-slot ring);
//  * fewer (N = 512: one tile per CU) -> ONE block of eight multiplying + four requesting waves (gemm_bf16_p192s_kernel:
//    FFN2 x1.15, dQKV x1.16, dFFN1 x1.16, output projection x1.15); two blocks per CU lose 3-10 % there.
#define g_p192_ring js2t_ctx_value(JS2T_CTX_GEMM_P192_RING)
template <int EPI>
int launch_bf16_p192_epi(const js2t_gemm_desc& d, hipStream_t s) {
  static int n_cu = 0;
  if (n_cu == 0) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_p192_kernel<EPI, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_bf16_p192_kernel<EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_bf16_p192s_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    int dev = 0, cu = 0;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu <= 0) {
      js2t_set_error("gemm p192 setup: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    n_cu = cu & ~7;  // a multiple of 8 keeps a block's tiles on one XCD's slice of the tile order
    if (n_cu == 0) n_cu = cu;
  }
  const int tm = cdiv(d.M, P_BM), tn = cdiv(d.N, 128);
  // the row statistics (rs_*) are written by the loader / consumer form only: its multiplying waves issue no LDS-DMA, so the
  // fences of the hand-over to the finishing wave wait for nothing but the wave's own stores
  // one block of twelve waves per CU while the tiles fit ONE round (<= one per CU); between one and one and a half tiles per CU that
  // form needs a second round for the few tiles left over (12288 < M <= 18432 rows at N = 512: the memory K | V input gradient on
  // the padded rows of a ragged batch, 284 tiles, took two tile times) - the two-block form has them all resident at once
  if (g_p192_ring == 4 || (EPI >= 0 && (EPI & (PE_STATS | PE_DOT))) || (g_p192_ring < 0 && tm * tn <= n_cu)) {
    const int grid = tm * tn < n_cu ? tm * tn : n_cu;
    hipLaunchKernelGGL((gemm_bf16_p192s_kernel<EPI>), dim3(grid), dim3(768), PS_LDS, s, d, tm, tn);
  } else if (g_p192_ring == 2 || (g_p192_ring < 0 && tm * tn > n_cu)) {
    const int grid = tm * tn < 2 * n_cu ? tm * tn : 2 * n_cu;
    hipLaunchKernelGGL((gemm_bf16_p192_kernel<EPI, 2>), dim3(grid), dim3(256), 2 * P_STAGE, s, d, tm, tn);
  } else {
    const int grid = tm * tn < n_cu ? tm * tn : n_cu;
    hipLaunchKernelGGL((gemm_bf16_p192_kernel<EPI, 3>), dim3(grid), dim3(256), P_LDS, s, d, tm, tn);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
// e4m3 operands: generic epilogue only (alpha carries the two per-tensor scales), ring depth by the same rule
int launch_fp8_p192(const js2t_gemm_desc& d, hipStream_t s) {
  static int n_cu = 0;
  if (n_cu == 0) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_p192_kernel<-1, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_bf16_p192_kernel<-1, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)gemm_bf16_p192s_kernel<-1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS);
    int dev = 0, cu = 0;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu <= 0) {
      js2t_set_error("gemm fp8 setup: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    n_cu = (cu & ~7) ? (cu & ~7) : cu;
  }
  const int tm = cdiv(d.M, P_BM), tn = cdiv(d.N, 128);
  if ((P192S_SCALED && g_p192_ring < 0) || ((P192S_SCALED || !d.c8) && (g_p192_ring == 4 || (g_p192_ring < 0 && 2 * tm * tn < 3 * n_cu)))) {
    // eight multiplying + four requesting waves.  With the block-scaled instruction (one v_mfma_scale_f32_16x16x128_f8f6f4 per
    // accumulator and stage) this form wins on every shape, also above 1.5 tiles per CU where the bf16 products prefer two blocks
    // per CU: FFN1 23.3 against 25.3 us, QKV 19.5 / 20.9, FFN2 15.5 / 23.1 (tools/fp8_gemm_bench.py); without it (-DJS2T_FP8_NO_SCALED)
    // the bf16 rule applies and the e4m3 second output lives in the ring forms only
    const int grid = tm * tn < n_cu ? tm * tn : n_cu;
    hipLaunchKernelGGL((gemm_bf16_p192s_kernel<-1, true>), dim3(grid), dim3(768), PS_LDS, s, d, tm, tn);
  } else if (g_p192_ring == 2 || (g_p192_ring < 0 && 2 * tm * tn >= 3 * n_cu)) {
    const int grid = tm * tn < 2 * n_cu ? tm * tn : 2 * n_cu;
    hipLaunchKernelGGL((gemm_bf16_p192_kernel<-1, 2, true>), dim3(grid), dim3(256), 2 * P_STAGE, s, d, tm, tn);
  } else {
    const int grid = tm * tn < n_cu ? tm * tn : n_cu;
    hipLaunchKernelGGL((gemm_bf16_p192_kernel<-1, 3, true>), dim3(grid), dim3(256), P_LDS, s, d, tm, tn);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
int launch_bf16_p192(const js2t_gemm_desc& d, hipStream_t s) {
  // the epilogue combinations of the Transformer train step get their own instantiation, anything else the generic one
  if (d.alpha == 1.f && !d.alpha_dev) {
    const int mask = (d.bias ? PE_BIAS : 0) | (d.act == JS2T_ACT_RELU ? PE_RELU : 0) | (d.dropout_p > 0.f ? PE_DROP : 0) |
                     (d.residual ? PE_RES : 0) | (d.gate ? PE_GATE : 0) | (d.ln_partial ? PE_LNF : 0) | (d.rs_partial ? PE_STATS : 0) |
                     (d.dot_partial ? PE_DOT : 0);
    // K <= 512 and many row strips per CU: the panel-resident kernel (gemm_panel.hip) streams A only
    const int rc = launch_bf16_pan96(d, mask, s);
    if (rc >= 0) return rc;
    switch (mask) {
      case PE_BIAS | PE_LNF: return launch_bf16_p192_epi<PE_BIAS | PE_LNF>(d, s);      // q/k/v projections on the raw residual stream
      case PE_BIAS | PE_RELU | PE_DROP | PE_LNF: return launch_bf16_p192_epi<PE_BIAS | PE_RELU | PE_DROP | PE_LNF>(d, s);  // FFN layer 1, same
      case PE_BIAS | PE_DROP | PE_RES | PE_STATS: return launch_bf16_p192_epi<PE_BIAS | PE_DROP | PE_RES | PE_STATS>(d, s);  // + next LN's row sums
      case PE_BIAS | PE_RELU | PE_LNF: return launch_bf16_p192_epi<PE_BIAS | PE_RELU | PE_LNF>(d, s);  // evaluation mode (no dropout)
      case PE_BIAS | PE_RES | PE_STATS: return launch_bf16_p192_epi<PE_BIAS | PE_RES | PE_STATS>(d, s);
      case 0: return launch_bf16_p192_epi<0>(d, s);                                    // input gradients
      case PE_BIAS: return launch_bf16_p192_epi<PE_BIAS>(d, s);                        // q/k/v projections
      case PE_BIAS | PE_RELU | PE_DROP: return launch_bf16_p192_epi<PE_BIAS | PE_RELU | PE_DROP>(d, s);  // FFN layer 1
      case PE_BIAS | PE_DROP | PE_RES: return launch_bf16_p192_epi<PE_BIAS | PE_DROP | PE_RES>(d, s);    // FFN layer 2, attention output
      case PE_GATE: return launch_bf16_p192_epi<PE_GATE>(d, s);                        // gradient through ReLU + dropout
      case PE_DOT: return launch_bf16_p192_epi<PE_DOT>(d, s);                          // attention output projection's input gradient + delta
      default: break;
    }
  }
  if (d.ln_partial || d.rs_partial || d.dot_partial) {
    js2t_set_error("gemm: LayerNorm fold: bias [+ ReLU [+ dropout]] with ln_partial, bias [+ dropout] + residual with rs_partial only; "
                   "dot_partial: plain epilogue only");
    return JS2T_ERR_INVALID;
  }
  return launch_bf16_p192_epi<-1>(d, s);
}

template <int BM, bool TA, bool TB, bool SPLITK, int NST = 2>
int launch_bf16_dma_bm(const js2t_gemm_desc& d, hipStream_t s) {
  constexpr int STAGE = ((BM == 128 || TA) ? 16384 : BM * 128) + 16384;
  constexpr int LDS = NST * STAGE > BM * 128 * 4 ? NST * STAGE : BM * 128 * 4;  // stages (the epilogue staging aliases them)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_dma_kernel<BM, TA, TB, SPLITK, NST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tm = cdiv(d.M, BM), tn = cdiv(d.N, F_BN);
  hipLaunchKernelGGL((gemm_bf16_dma_kernel<BM, TA, TB, SPLITK, NST>), dim3(tm * tn, d.batch, SPLITK ? d.split_k : 1), dim3(256), LDS,
                     s, d, tm, tn, (float*)d.C, d.split_k);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
template <bool TA, bool TB, bool SPLITK>
int launch_bf16_dma(const js2t_gemm_desc& d, hipStream_t s) {
  if constexpr (!TA && !SPLITK) {
    // few 128x128 tiles (N = 512 layers, decoder-sized M): halve the row tile to put ~2x more blocks on the 256 CUs
    const int64_t tiles = (int64_t)cdiv(d.M, 128) * cdiv(d.N, F_BN) * d.batch;
    if (tiles < 512 && d.M > 64) {
      // fewer 64-row tiles than CUs: one block per CU at most, so trade the idle LDS for a 4-deep DMA ring
      if ((int64_t)cdiv(d.M, 64) * cdiv(d.N, F_BN) * d.batch <= 256 && d.K > 128)
        return launch_bf16_dma_bm<64, TA, TB, SPLITK, 4>(d, s);
      return launch_bf16_dma_bm<64, TA, TB, SPLITK>(d, s);
    }
  }
  return launch_bf16_dma_bm<128, TA, TB, SPLITK>(d, s);
}

template <bool TA, bool TB, bool SPLITK>
int launch_bf16_impl(const js2t_gemm_desc& d, hipStream_t s) {
  constexpr int A_BYTES = TA ? TR_TILE_BYTES : KC_TILE_BYTES;
  constexpr int B_BYTES = TB ? TR_TILE_BYTES : KC_TILE_BYTES;
  constexpr int LDS = 2 * (A_BYTES + B_BYTES);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_kernel<TA, TB, SPLITK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tm = cdiv(d.M, F_BM), tn = cdiv(d.N, F_BN);
  hipLaunchKernelGGL((gemm_bf16_kernel<TA, TB, SPLITK>), dim3(tm * tn, d.batch, SPLITK ? d.split_k : 1), dim3(256), LDS, s, d,
                     tm, tn, (float*)d.C, d.split_k);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
template <bool TA, bool TB>
int launch_bf16(const js2t_gemm_desc& d, hipStream_t s) {
  if (!TA && !TB && !g_force_regstage && !d.dot_partial && w256_eligible(d)) return launch_bf16_w256(d, s);
  if (!TA && !TB && !g_force_regstage && !g_force_w256 && p192_eligible(d)) return launch_bf16_p192(d, s);
  if (!d.conv && !g_force_regstage)
    return d.split_k > 1 ? launch_bf16_dma<TA, TB, true>(d, s) : launch_bf16_dma<TA, TB, false>(d, s);
  return d.split_k > 1 ? launch_bf16_impl<TA, TB, true>(d, s) : launch_bf16_impl<TA, TB, false>(d, s);
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" void js2t_gemm_force_regstage(int on) { g_force_regstage = on != 0; }
extern "C" void js2t_gemm_force_w256(int on) { g_force_w256 = on != 0; }
#ifdef JS2T_P192_PROF
extern "C" int js2t_debug_p192_prof(unsigned long long* out8) {
  return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_p192_prof), 64);
}
extern "C" int js2t_debug_p192_prof2(unsigned long long* out8, int reset) {
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_p192_prof2), z, 64);
  }
  return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_p192_prof2), 64);
}
#endif
extern "C" void js2t_gemm_p192_ring(int nst) { js2t_ctx_override(JS2T_CTX_GEMM_P192_RING, (nst >= 2 && nst <= 4) ? nst : -1); }
extern "C" void js2t_gemm_p192_mode(int mode) { js2t_ctx_override(JS2T_CTX_GEMM_P192_MODE, mode < 0 ? -1 : (mode > 2 ? 1 : mode)); }

template <bool SPLITK>
static int launch_grouped_tt(const js2t_gemm_desc& d, const GemmGroup& grp, int count, hipStream_t s) {
  constexpr int LDS = 65536;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_dma_grouped_kernel<128, true, true, SPLITK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tm = cdiv(d.M, 128), tn = cdiv(d.N, F_BN);
  hipLaunchKernelGGL((gemm_bf16_dma_grouped_kernel<128, true, true, SPLITK>), dim3(tm * tn * count * (SPLITK ? d.split_k : 1)), dim3(256),
                     LDS, s, d, grp, tm, tn, d.split_k);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// the persistent reduction-major kernel takes a grouped launch when it can fill the chip with 192x128 tiles
static bool p192t_eligible(const js2t_gemm_desc& d, int count) {
  if (g_p192_mode == 0 || g_p192_mode == 2 || d.split_k > 1 || d.dtype_c != JS2T_F32) return false;
  if ((d.N & 127) || (d.M & 7) || d.K < 129 || (d.ldc & 0) != 0) return false;
  // Measured in the train step (real activations) the fixed two-stage kernel wins on every weight-gradient group:
  // 839 TFLOP/s over FFN1 / attention-output / key-value launches against 773 for this kernel on 16 x dW[1536,512]
  // (exact fit: 512 tiles, no partial row tile) and 735 on dW[512,2048] + dW[1536,512]; synthetic normal operands
  // favour it on the exact fit only (806 vs 698).  It stays selectable (mode 1) and tested, not the default.
  if (g_p192_mode != 1) return false;
  return true;
}
template <bool RS>
static int launch_grouped_p192t(const js2t_gemm_desc& d, const GemmGroup& grp, int count, hipStream_t s) {
  static int n_cu = 0;
  if (n_cu == 0) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_p192t_kernel<RS>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    int dev = 0, cu = 0;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu <= 0) {
      js2t_set_error("gemm p192t setup: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    n_cu = (cu & ~7) ? (cu & ~7) : cu;
  }
  const int tm = cdiv(d.M, P_BM), tn = d.N >> 7;
  const int total = tm * tn * count, grid = total < n_cu ? total : n_cu;
  hipLaunchKernelGGL(gemm_bf16_p192t_kernel<RS>, dim3(grid), dim3(256), P_LDS, s, d, grp, tm, tn, count);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

// -1 (default): the grouped launches that fill the chip with 256x128 tiles over a long reduction; 0: never; 1: whenever the shape allows
#define g_wg256_mode js2t_ctx_value(JS2T_CTX_GEMM_WG256_MODE)
extern "C" void js2t_gemm_wg256_mode(int mode) { js2t_ctx_override(JS2T_CTX_GEMM_WG256_MODE, mode < 0 ? -1 : (mode > 1 ? 1 : mode)); }
static bool wg256_eligible(const js2t_gemm_desc& d, int count) {
  if (g_wg256_mode == 0 || d.dtype_c != JS2T_F32 || (d.split_k > 1 && d.sumsq_partial)) return false;
  if ((d.M & 255) || (d.N & 127) || (d.ldc & 3) || d.K < 192 * d.split_k || d.split_k > ((d.K + 63) >> 6) / 4) return false;
  if (g_wg256_mode == 1) return true;
  const int64_t blocks = (int64_t)(d.M >> 8) * (d.N >> 7) * count * d.split_k;
  return d.K >= 1024 * d.split_k && blocks >= 160;  // one block per CU: below that most of the chip idles, the 128x128 tiles spread further
}
template <bool RS, bool SPLITK>
static int launch_grouped_wg256(const js2t_gemm_desc& d, const GemmGroup& grp, int count, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)(gemm_bf16_wg256_kernel<RS, SPLITK>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS);
    if (e != hipSuccess) {
      js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int tm = d.M >> 8, tn = d.N >> 7;
  hipLaunchKernelGGL((gemm_bf16_wg256_kernel<RS, SPLITK>), dim3(tm * tn * count * (SPLITK ? d.split_k : 1)), dim3(768), WG_LDS, s, d, grp, tm, tn,
                     d.split_k);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_gemm_grouped(const js2t_gemm_desc* dp, int32_t count, const void* const* A, const void* const* B, void* const* C,
                                 float* const* a_rowsum, js2t_stream stream) {
  JS2T_CHECK(dp != nullptr && count >= 0, "gemm_grouped: bad arguments");
  if (count == 0) return JS2T_OK;
  JS2T_CHECK(A && B && C, "gemm_grouped: null pointer tables");
  js2t_gemm_desc d = *dp;
  JS2T_CHECK(d.M > 0 && d.N > 0 && d.K > 0, "gemm_grouped: empty product");
  JS2T_CHECK(d.dtype_ab == JS2T_BF16 && d.trans_a && d.trans_b && !d.conv && d.batch == 1,
             "gemm_grouped: bf16 reduction-major operands only (trans_a = trans_b = 1, batch = 1)");
  JS2T_CHECK((d.lda % 8 == 0) && (d.ldb % 8 == 0), "gemm_grouped: leading dimensions must be multiples of 8");
  JS2T_CHECK(d.dtype_c == JS2T_F32 || d.dtype_c == JS2T_BF16, "gemm_grouped: bad dtype_c");
  JS2T_CHECK(!d.bias && d.act == JS2T_ACT_NONE && !d.preact && d.dropout_p == 0.f && !d.residual && !d.gate && !d.alpha_dev,
             "gemm_grouped: plain epilogue only (alpha, beta)");
  if (d.split_k < 1) d.split_k = 1;
  JS2T_CHECK(d.split_k == 1 || (d.dtype_c == JS2T_F32 && d.beta == 0.f), "gemm_grouped: split_k needs f32 C and beta = 0");
  if (g_js2t_deterministic && d.split_k > 1) d.split_k = 1, d.beta = 1.f;  // K slices summed by atomics -> one slice added onto C
  if (d.sumsq_partial) {
    // every tile must take the epilogue's row-segment path: whole 128-column tiles, 16-byte aligned f32 rows
    JS2T_CHECK(d.split_k == 1 && d.dtype_c == JS2T_F32 && (d.N & 127) == 0 && (d.ldc & 3) == 0,
               "gemm_grouped: sumsq_partial needs f32 C, no split-K, N % 128 == 0, ldc % 4 == 0");
    for (int i = 0; i < count; ++i) JS2T_CHECK(aligned16(C[i]), "gemm_grouped: sumsq_partial needs 16-byte aligned C");
  }
  for (int base = 0; base < count; base += JS2T_GEMM_GROUP_MAX) {
    const int n = count - base < JS2T_GEMM_GROUP_MAX ? count - base : JS2T_GEMM_GROUP_MAX;
    GemmGroup grp;
    for (int i = 0; i < JS2T_GEMM_GROUP_MAX; ++i) {
      const int j = base + (i < n ? i : 0);
      JS2T_CHECK(A[j] && B[j] && C[j] && aligned16(A[j]) && aligned16(B[j]), "gemm_grouped: operands must be non-null, 16-byte aligned");
      grp.A[i] = A[j], grp.B[i] = B[j], grp.C[i] = C[j];
      grp.rowsum[i] = a_rowsum ? a_rowsum[j] : nullptr;
    }
    d.A = grp.A[0], d.B = grp.B[0], d.C = grp.C[0];
    int rc;
    bool c16 = true;  // (the 256x128 kernel stores 16-byte pieces)
    for (int i = 0; i < n; ++i) c16 = c16 && aligned16(grp.C[i]);
    if (c16 && wg256_eligible(d, n)) {
      if (d.split_k > 1)
        rc = a_rowsum ? launch_grouped_wg256<true, true>(d, grp, n, (hipStream_t)stream) : launch_grouped_wg256<false, true>(d, grp, n, (hipStream_t)stream);
      else
        rc = a_rowsum ? launch_grouped_wg256<true, false>(d, grp, n, (hipStream_t)stream) : launch_grouped_wg256<false, false>(d, grp, n, (hipStream_t)stream);
    } else if (!d.sumsq_partial && p192t_eligible(d, n)) {
      rc = a_rowsum ? launch_grouped_p192t<true>(d, grp, n, (hipStream_t)stream) : launch_grouped_p192t<false>(d, grp, n, (hipStream_t)stream);
    } else {
      rc = d.split_k > 1 ? launch_grouped_tt<true>(d, grp, n, (hipStream_t)stream) : launch_grouped_tt<false>(d, grp, n, (hipStream_t)stream);
    }
    if (rc != JS2T_OK) return rc;
    if (d.sumsq_partial) d.sumsq_partial += (int64_t)cdiv(d.M, 128) * cdiv(d.N, F_BN) * n;
  }
  return JS2T_OK;
}

extern "C" int64_t js2t_gemm_grouped_blocks(int32_t M, int32_t N, int32_t count) {
  return (int64_t)cdiv(M, 128) * cdiv(N, F_BN) * count;
}

extern "C" int js2t_gemm(const js2t_gemm_desc* dp, js2t_stream stream) {
  JS2T_CHECK(dp != nullptr, "gemm: null descriptor");
  js2t_gemm_desc d = *dp;
  JS2T_CHECK(!d.sumsq_partial, "gemm: sumsq_partial is taken by js2t_gemm_grouped only");
  hipStream_t s = (hipStream_t)stream;
  JS2T_CHECK(d.M >= 0 && d.N >= 0 && d.K >= 0 && d.batch >= 0, "gemm: negative size");
  if (d.M == 0 || d.N == 0 || d.batch == 0) return JS2T_OK;
  JS2T_CHECK(d.A && d.B && (d.C || (d.c8 && d.dtype_ab == JS2T_FP8_E4M3)), "gemm: null operand");
  JS2T_CHECK(!d.c8 || (d.dtype_ab == JS2T_FP8_E4M3 && d.c8_state && d.ldc8 >= d.N && (d.ldc8 & 7) == 0 && (((uintptr_t)d.c8) & 7) == 0 &&
                       (((uintptr_t)d.c8_state) & 15) == 0),
             "gemm: c8 (e4m3 second output) needs e4m3 operands, c8_state (16-byte aligned) and 8-byte aligned rows");
  JS2T_CHECK(d.batch_inner >= 1, "gemm: batch_inner must be >= 1");
  JS2T_CHECK(d.dtype_ab == JS2T_F32 || d.dtype_ab == JS2T_BF16 || d.dtype_ab == JS2T_FP8_E4M3, "gemm: bad dtype_ab");
  JS2T_CHECK(d.dtype_c == JS2T_F32 || d.dtype_c == JS2T_BF16, "gemm: bad dtype_c");
  JS2T_CHECK(d.dropout_p >= 0.f && d.dropout_p < 1.f, "gemm: dropout_p out of range");
  JS2T_CHECK(d.dropout_p == 0.f || d.rng_state, "gemm: dropout needs rng_state");
  JS2T_CHECK(!d.residual || d.ldr > 0, "gemm: residual needs ldr");
  JS2T_CHECK(!d.gate || d.ldg > 0, "gemm: gate needs ldg");
  JS2T_CHECK(d.batch <= 65535, "gemm: batch too large");
  if (d.conv) {
    JS2T_CHECK(d.conv_c > 0 && d.conv_tin > 0 && d.conv_tout > 0 && d.conv_stride > 0, "gemm: bad conv geometry");
  }
  JS2T_CHECK(!(d.residual || d.gate) || d.batch == 1, "gemm: residual / gate need batch == 1");
  JS2T_CHECK(!d.a_rowsum || (d.batch == 1 && !d.conv && !g_force_regstage && d.trans_a && d.trans_b),
             "gemm: a_rowsum needs trans_a = trans_b = 1, batch == 1 and the LDS-DMA kernel");
  if (d.split_k < 1) d.split_k = 1;
  if (d.split_k > 1) {
    JS2T_CHECK(d.dtype_c == JS2T_F32 && !d.bias && d.act == JS2T_ACT_NONE && !d.preact && d.dropout_p == 0.f && !d.residual &&
                   !d.gate && d.beta == 0.f,
               "gemm: split_k needs an f32 C and a plain epilogue (C must be zero-filled by the caller)");
    if (g_js2t_deterministic) d.split_k = 1, d.beta = 1.f;  // one slice, added onto C: what the atomics of the slices do, in one order
  }
  if (d.ln_partial || d.rs_partial) {
    // the fold lives in the register-direct epilogues of the k-contiguous bf16 kernels: the persistent 192x128 kernel (three
    // forms) and, for products with too few tiles for it, the 64 / 128-row tile kernel
    JS2T_CHECK(d.dtype_ab == JS2T_BF16 && d.dtype_c == JS2T_BF16 && !d.trans_a && !d.trans_b && !d.conv && d.split_k == 1 && d.batch == 1 &&
                   !d.preact && d.beta == 0.f && !d.a_rowsum && !(d.residual && d.gate) && (d.act == JS2T_ACT_NONE || d.act == JS2T_ACT_RELU) &&
                   !g_force_regstage && !g_force_w256 && d.alpha == 1.f && !d.alpha_dev && d.bias,
               "gemm: ln_partial / rs_partial need a plain k-contiguous bf16 product with a bf16 result, a bias and alpha == 1");
    JS2T_CHECK((d.N & 127) == 0 && (d.K & 7) == 0 && d.K >= 64 && (d.lda & 7) == 0 && (d.ldb & 7) == 0 && (d.ldc & 7) == 0 && aligned16(d.A) &&
                   aligned16(d.B) && aligned16(d.C) && aligned16(d.bias) && (!d.residual || ((d.ldr & 7) == 0 && aligned16(d.residual))) &&
                   (!d.gate || ((d.ldg & 7) == 0 && aligned16(d.gate))),
               "gemm: ln_partial / rs_partial need N % 128 == 0 and 16-byte aligned rows");
    const int fmask = (d.act == JS2T_ACT_RELU ? 2 : 0) | (d.dropout_p > 0.f ? 4 : 0) | (d.residual ? 8 : 0) | (d.gate ? 16 : 0);
    JS2T_CHECK(!d.ln_partial || fmask == 0 || fmask == 2 || fmask == 6, "gemm: ln_partial: epilogue bias [+ ReLU [+ dropout]] only");
    JS2T_CHECK(!d.rs_partial || fmask == 8 || fmask == 12, "gemm: rs_partial: epilogue bias [+ dropout] + residual only");
    JS2T_CHECK(!d.ln_partial || (d.K == 64 * LNF_GROUPS && aligned16(d.ln_partial) && d.ln_eps > 0.f && (!d.ln_mean == !d.ln_rstd)),
               "gemm: ln_partial: K must be 512 (eight 64-column groups), 16-byte aligned partial sums, ln_eps > 0, ln_mean / ln_rstd both or neither");
    JS2T_CHECK(!d.rs_partial || (d.N == 64 * LNF_GROUPS && aligned16(d.rs_partial)), "gemm: rs_partial: N must be 512, 16-byte aligned partial sums");
  }
  if (d.dot_partial) {
    // written by the register-direct epilogues of the k-contiguous bf16 kernels (loader / consumer form of the persistent kernel,
    // 64 / 128-row tile kernel): every tile must take them
    JS2T_CHECK(d.dot_src && d.dtype_ab == JS2T_BF16 && d.dtype_c == JS2T_BF16 && !d.trans_a && !d.trans_b && !d.conv && d.split_k == 1 &&
                   d.batch == 1 && !d.preact && d.beta == 0.f && !d.a_rowsum && !d.bias && d.act == JS2T_ACT_NONE && d.dropout_p == 0.f &&
                   !d.residual && !d.gate && d.alpha == 1.f && !d.alpha_dev && !d.ln_partial && !d.rs_partial && !g_force_regstage && !g_force_w256,
               "gemm: dot_partial needs a plain k-contiguous bf16 product with a bf16 result (no bias / activation / dropout / residual)");
    JS2T_CHECK((d.N & 127) == 0 && (d.K & 7) == 0 && (d.lda & 7) == 0 && (d.ldb & 7) == 0 && (d.ldc & 7) == 0 && (d.ld_dot & 7) == 0 &&
                   aligned16(d.A) && aligned16(d.B) && aligned16(d.C) && aligned16(d.dot_src) && d.ld_dot >= d.N,
               "gemm: dot_partial needs N % 128 == 0 and 16-byte aligned rows of A, B, C and dot_src");
  }
  if (d.dtype_ab == JS2T_FP8_E4M3) {
    // e4m3 x e4m3 -> f32 accumulate -> bf16: the persistent 192x128 kernel only (k-contiguous operands, 16-byte rows)
    JS2T_CHECK(!d.trans_a && !d.trans_b && !d.conv && d.split_k == 1 && d.batch == 1 && d.dtype_c == JS2T_BF16 && !d.preact &&
                   d.beta == 0.f && !d.a_rowsum && !(d.residual && d.gate) && (d.act == JS2T_ACT_NONE || d.act == JS2T_ACT_RELU),
               "gemm fp8: plain k-contiguous products with a bf16 result only");
    JS2T_CHECK((d.N & 7) == 0 && d.N >= 128 && (d.K & 15) == 0 && d.K >= 128 && (d.lda & 15) == 0 && (d.ldb & 15) == 0 && aligned16(d.A) &&
                   aligned16(d.B) && (!d.C || aligned16(d.C)) && (d.ldc & 7) == 0,
               "gemm fp8: N % 8 == 0, N >= 128, K % 16 == 0, K >= 128, 16-byte aligned rows");
    JS2T_CHECK(!d.residual || ((d.ldr & 7) == 0 && aligned16(d.residual)), "gemm fp8: misaligned residual");
    JS2T_CHECK(!d.gate || ((d.ldg & 7) == 0 && aligned16(d.gate)), "gemm fp8: misaligned gate");
    return launch_fp8_p192(d, s);
  }
  bool fast = d.dtype_ab == JS2T_BF16 && aligned16(d.A) && aligned16(d.B) && (d.lda % 8 == 0) && (d.ldb % 8 == 0) &&
              (d.a_stride_o % 8 == 0) && (d.a_stride_i % 8 == 0) && (d.b_stride_o % 8 == 0) && (d.b_stride_i % 8 == 0) &&
              (!d.conv || d.conv_c % 8 == 0) && d.K > 0;
  if (fast) {
    if (!d.trans_a && !d.trans_b) return launch_bf16<false, false>(d, s);
    if (!d.trans_a && d.trans_b) return launch_bf16<false, true>(d, s);
    if (d.trans_a && !d.trans_b) return launch_bf16<true, false>(d, s);
    return launch_bf16<true, true>(d, s);
  }
  JS2T_CHECK(!d.a_rowsum, "gemm: a_rowsum needs bf16 operands on the LDS-DMA path (16-byte aligned, ld % 8 == 0)");
  d.split_k = 1;  // the generic kernel always reduces the whole K range (C stays pre-zeroed + one plain store)
  const int tm = cdiv(d.M, G_BM), tn = cdiv(d.N, G_BN);
  hipLaunchKernelGGL(gemm_generic_kernel, dim3(tm * tn, d.batch), dim3(256), 0, s, d, tm, tn);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
