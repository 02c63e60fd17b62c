// EXPERIMENT, not part of libjoeys2t_hip.so (round 3; profiles/README.md "weight-stationary K = 512 product"): correct - bit-identical
// to js2t_gemm on every shape tried - and slower than the persistent 192x128 kernel it was meant to replace (FFN1 12000 x 2048 x 512:
// 40.4 us against 36.6).  Kept as a record of the design and of where its time went; to build it, copy it into joeys2t_amd/csrc/
// (the library build picks every .hip there up) and call js2t_debug_gemm_ws512 (tools/gemm_ws_probe.py).
//
// Weight-stationary product for K = 512: C[M, N] = epilogue(X[M, 512] W[N, 512]^T), bf16 in, bf16 out.
//
// The persistent 192x128 kernel (gemm.hip) pulls 40 KB from L2 into LDS per 3.1 MFLOP and sits at the L2 -> LDS delivery
// rate on the train step's K = 512 products (QKV, FFN1, out-projection of transformer_layers.py:75-107,147-153 and the
// input gradients of the same width): 8 K steps per tile, every tile loads its own slice of W again.  Here a block keeps a
// 256-column panel of W - all 512 of its k - in REGISTERS for its whole life (64 columns per wave = 64 MFMA fragments = 256
// dwords per lane, which is what one wave per SIMD is allowed) and streams rows of X past it: 32 rows x 512 k = 32 KB per
// 8.4 MFLOP through a four-slot LDS ring, 0.3x the bytes per flop, one LDS fragment read per four MFMAs.
// One block per CU; block = (column panel, row range); panels of one XCD share their W lines in that XCD's L2.
#include "common.hpp"

namespace {

typedef __attribute__((address_space(1))) const void g_cvoid;
typedef __attribute__((address_space(3))) void l_void;

constexpr int WS_K = 512, WS_KS = WS_K / 32;      // 16 MFMA k-steps
constexpr int WS_ROWS = 32;                        // rows of X per ring slot
constexpr int WS_SLOT = WS_ROWS * WS_K * 2;        // 32 KB
constexpr int WS_NSLOT = 4, WS_AHEAD = WS_NSLOT - 1;
constexpr int WS_LDS = WS_NSLOT * WS_SLOT;         // 128 KB
constexpr int WS_PANEL = 256;                      // columns per block, 64 per wave

constexpr int WE_BIAS = 1, WE_RELU = 2;

// slot image: 8 k-blocks of [32 rows][64 k] bf16 (4 KB each), 16-byte granule g of row m at position g ^ (m & 7): the
// fragment reads below (16 consecutive rows, one granule column) are conflict-free.  A wave requests 8 of the 32 one-KB
// pieces of a chunk; its lanes keep running source pointers (src[q] walks down X by 32 rows per chunk).
__device__ __forceinline__ void ws_piece(int w, int q, int lane, int& kb, int& m, int& g) {
  const int piece = w * 8 + q;  // k-block kb = piece / 4, rows 8 * (piece % 4) ..
  kb = piece >> 2, m = ((piece & 3) << 3) + (lane >> 3), g = (lane & 7) ^ (m & 7);
}
__device__ __forceinline__ void ws_request(const uint16_t* const (&src)[8], unsigned char* slot, int w) {
#pragma unroll
  for (int q = 0; q < 8; ++q) __builtin_amdgcn_global_load_lds((g_cvoid*)src[q], (l_void*)(slot + (w * 8 + q) * 1024), 16, 0, 0);
}
// the chunk that holds the last rows of X: rows past M - 1 are read from row M - 1 (never stored)
__device__ __forceinline__ void ws_request_clamped(const uint16_t* __restrict__ X, int64_t ldx, int row0, int M, unsigned char* slot, int w,
                                                   int lane) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    int kb, m, g;
    ws_piece(w, q, lane, kb, m, g);
    const uint16_t* p = X + (int64_t)min(row0 + m, M - 1) * ldx + kb * 64 + g * 8;
    __builtin_amdgcn_global_load_lds((g_cvoid*)p, (l_void*)(slot + (w * 8 + q) * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8_t ws_xfrag(const unsigned char* slot, int mt, int ks, int lane) {
  const int row = 16 * mt + (lane & 15), c = (ks & 1) * 4 + (lane >> 4);
  return *(const bf16x8_t*)(slot + (ks >> 1) * 4096 + row * 128 + ((c ^ (row & 7)) << 4));
}

#ifdef JS2T_WS_PROF
__device__ unsigned long long g_ws_prof[8];
#define WS_T(i)                                                 \
  do {                                                          \
    const unsigned long long c_ = __builtin_readcyclecounter(); \
    prof_[i] += c_ - last_;                                     \
    last_ = c_;                                                 \
  } while (0)
#else
#define WS_T(i)
#endif

// bias / ReLU / bf16 of one lane's 16 consecutive columns of row block mt -> two 16-byte stores
template <int EPI>
__device__ __forceinline__ void ws_store_rows(const f32x4_t (&acc)[2][4], const float (&bias_r)[16], uint16_t* __restrict__ dst0, int64_t ldc16,
                                              int rows_left) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    uint32_t pk[8];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v0 = acc[mt][ct][2 * h] + bias_r[4 * ct + 2 * h], v1 = acc[mt][ct][2 * h + 1] + bias_r[4 * ct + 2 * h + 1];
        if (EPI & WE_RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
        pk[2 * ct + h] = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
      }
    if (16 * mt < rows_left) {  // rows_left: 32 + for a whole chunk (the compiler drops the test), the lane's own count in the last one
      uint16_t* dst = dst0 + mt * ldc16;
      *(uint4*)dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      *(uint4*)(dst + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
  }
}

// One chunk: its 128 MFMAs in 16 k-steps, and inside every k-step - fenced, so that the scheduler spreads them no further - a
// sixteenth of what else has to happen: the fragment reads of the NEXT k-step, one packed pair of the epilogue of the chunk
// before (whose sums sit in the other accumulator set), every other step one of this wave's eight requests for the chunk
// three ahead.  One wave per SIMD has nobody else to fill the MFMA shadows with.
template <int EPI, bool HAS_PREV, bool REQ>
__device__ __forceinline__ void ws_chunk(const unsigned char* slot, const bf16x8_t (&wf)[4][WS_KS], f32x4_t (&acc)[2][4],
                                         const f32x4_t (&prev)[2][4], const float (&bias_r)[16], uint16_t* __restrict__ prev_dst, int64_t ldc16,
                                         int lane, const uint16_t* req_base, const int64_t (&req_off)[8], unsigned char* req_slot, int w) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[mt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t xf[2][2];
  xf[0][0] = ws_xfrag(slot, 0, 0, lane), xf[0][1] = ws_xfrag(slot, 1, 0, lane);
  uint32_t pk[8];
#pragma unroll
  for (int ks = 0; ks < WS_KS; ++ks) {
    const int cur = ks & 1;
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < WS_KS) xf[cur ^ 1][0] = ws_xfrag(slot, 0, ks + 1, lane), xf[cur ^ 1][1] = ws_xfrag(slot, 1, ks + 1, lane);
    if (REQ && (ks & 1)) {
      const int q = ks >> 1;
      __builtin_amdgcn_global_load_lds((g_cvoid*)(req_base + req_off[q]), (l_void*)(req_slot + (w * 8 + q) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][ks], xf[cur][mt], acc[mt][ct], 0, 0, 0);
    if (HAS_PREV) {  // k-steps 0-7: row block 0, 8-15: row block 1; one (column block, half) per step, the two stores with the last
      const int mt = ks >> 3, ct = (ks >> 1) & 3, h = ks & 1;
      float v0 = prev[mt][ct][2 * h] + bias_r[4 * ct + 2 * h], v1 = prev[mt][ct][2 * h + 1] + bias_r[4 * ct + 2 * h + 1];
      if (EPI & WE_RELU) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
      pk[2 * ct + h] = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
      if ((ks & 7) == 7) {
        uint16_t* dst = prev_dst + mt * ldc16;
        *(uint4*)dst = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *(uint4*)(dst + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_ws512_kernel(const uint16_t* __restrict__ X, int64_t ldx, const uint16_t* __restrict__ W,
                                                           int64_t ldw, uint16_t* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ bias, int M, int N, int ranges, int rows_per) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, g = lane >> 4, j = lane & 15;
  // work item = (row range, column panel), panel fastest: the panels of one row range sit on one XCD (xcd_remap hands an XCD a
  // contiguous run of items), so a chunk of X comes through the fabric once and the XCD's L2 serves the other panels
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int panels = N / WS_PANEL;
  const int range = item / panels, panel = item - range * panels;
  const int r_lo = range * rows_per, r_hi = min(M, r_lo + rows_per);
  if (r_lo >= r_hi) return;
#ifdef JS2T_WS_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  const int nch = (r_hi - r_lo + WS_ROWS - 1) / WS_ROWS;
  const int n0 = panel * WS_PANEL + 64 * w;  // this wave's 64 columns
  const bool ragged = r_lo + nch * WS_ROWS > M;  // the last chunk reaches past the last row of X (only the last range's can)

  // this wave's eight one-KB pieces of a chunk: lane-dependent base (row lane / 8, swizzled granule), piece-dependent uniform offset
  int64_t req_off[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) req_off[q] = (int64_t)(8 * ((w * 8 + q) & 3)) * ldx + ((w * 8 + q) >> 2) * 64;
  const uint16_t* req_base = X + (int64_t)(r_lo + (lane >> 3)) * ldx + (((lane & 7) ^ ((lane >> 3) & 7)) << 3);
  const int64_t step = (int64_t)WS_ROWS * ldx;
  // the first rows of X are on their way while the panel of W arrives (chunks past the end: the last one again - the number of
  // requests per iteration is what the counted wait below relies on)
#pragma unroll
  for (int c = 0; c < WS_AHEAD; ++c) {
    const int cc = min(c, nch - 1);
    if (ragged && cc == nch - 1) ws_request_clamped(X, ldx, r_lo + cc * WS_ROWS, M, smem + c * WS_SLOT, w, lane);
    else {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((g_cvoid*)(req_base + cc * step + req_off[q]), (l_void*)(smem + c * WS_SLOT + (w * 8 + q) * 1024), 16, 0, 0);
    }
  }

  // W fragments: MFMA operand row i = lane & 15 of column block ct is column n0 + 16 * (i / 4) + 4 * ct + (i % 4), so that the
  // result rows 4g .. 4g+3 of the four column blocks are 16 CONSECUTIVE output columns of lane group g
  bf16x8_t wf[4][WS_KS];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int n = n0 + 16 * (j >> 2) + 4 * ct + (j & 3);
    const uint16_t* wr = W + (int64_t)n * ldw + 8 * g;
#pragma unroll
    for (int ks = 0; ks < WS_KS; ++ks) wf[ct][ks] = *(const bf16x8_t*)(wr + 32 * ks);
  }
  float bias_r[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) bias_r[e] = (EPI & WE_BIAS) ? bias[n0 + 16 * g + e] : 0.f;
  // lane (g, j): rows 16 mt + j of a chunk, columns n0 + 16g .. +15 (acc[mt][ct][r] is column 16g + 4ct + r)
  uint16_t* dst = C + (int64_t)(r_lo + j) * ldc + n0 + 16 * g;  // of the chunk whose epilogue comes next
  const int64_t ldc16 = 16 * ldc, ldc32 = 32 * ldc;

  f32x4_t accA[2][4], accB[2][4];
  WS_T(0);  // prologue: first requests, W panel, bias
  // Per iteration a wave issues 8 requests and (from the second on) 4 stores; its pieces of chunk c were requested three
  // iterations ago, so they have landed when at most the 2 x 8 requests and 2 x 4 stores issued since are in flight (the
  // queue retires in order; the stores of iteration c - 3 may sit on either side of the requests: 24 covers both).
#define WS_ITER(ACC, PREV, HAS_PREV)                                                                          \
  do {                                                                                                        \
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                                                         \
    WS_T(1);                                                                                                  \
    __builtin_amdgcn_s_barrier(); /* raw: __syncthreads() would wait for EVERY request in flight.  All pieces of chunk c are in; every wave is through with chunk c - 1, whose slot the request takes */ \
    WS_T(2);                                                                                                  \
    const int cn = min(c + WS_AHEAD, nch - 1);                                                                \
    unsigned char* rs = smem + ((c + WS_AHEAD) % WS_NSLOT) * WS_SLOT;                                         \
    if (ragged && cn == nch - 1) {                                                                            \
      ws_request_clamped(X, ldx, r_lo + cn * WS_ROWS, M, rs, w, lane);                                        \
      ws_chunk<EPI, HAS_PREV, false>(smem + (c % WS_NSLOT) * WS_SLOT, wf, ACC, PREV, bias_r, dst, ldc16, lane, nullptr, req_off, rs, w); \
    } else {                                                                                                  \
      ws_chunk<EPI, HAS_PREV, true>(smem + (c % WS_NSLOT) * WS_SLOT, wf, ACC, PREV, bias_r, dst, ldc16, lane, req_base + cn * step, req_off, rs, w); \
    }                                                                                                         \
    if (HAS_PREV) dst += ldc32;                                                                               \
    WS_T(4);                                                                                                  \
  } while (0)
  int c = 0;
  WS_ITER(accA, accB, false);
  for (c = 1; c + 1 < nch; c += 2) {
    WS_ITER(accB, accA, true);
    ++c;
    WS_ITER(accA, accB, true);
    --c;
  }
  if (c < nch) {  // an even number of chunks: the last one goes to B
    WS_ITER(accB, accA, true);
    ws_store_rows<EPI>(accB, bias_r, dst, ldc16, r_hi - (r_lo + (nch - 1) * WS_ROWS) - j);
  } else {
    ws_store_rows<EPI>(accA, bias_r, dst, ldc16, r_hi - (r_lo + (nch - 1) * WS_ROWS) - j);
  }
#undef WS_ITER
  WS_T(5);
#ifdef JS2T_WS_PROF
  if (blockIdx.x == 0 && t == 0) {
    for (int i = 0; i < 6; ++i) g_ws_prof[i] = prof_[i];
    g_ws_prof[6] = nch;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests past the end of the range still write LDS: not while the block retires
}

}  // namespace

#ifdef JS2T_WS_PROF
extern "C" int js2t_debug_ws_prof(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_prof), sizeof(g_ws_prof)) == hipSuccess ? 0 : -1;
}
#endif

// Debug / bring-up entry: X [M, 512] bf16 (ldx), W [N, 512] bf16 (ldw), C [M, N] bf16 (ldc), N % 256 == 0, optional f32 bias and ReLU.
extern "C" int js2t_debug_gemm_ws512(const void* X, int64_t ldx, const void* W, int64_t ldw, void* C, int64_t ldc, const float* bias,
                                     int32_t relu, int32_t M, int32_t N, js2t_stream stream) {
  JS2T_CHECK(X && W && C && M > 0 && N > 0 && N % WS_PANEL == 0, "gemm_ws512: N must be a multiple of 256");
  JS2T_CHECK((((uintptr_t)X | (uintptr_t)W | (uintptr_t)C) & 15) == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0,
             "gemm_ws512: 16-byte aligned rows");
  const int panels = N / WS_PANEL;
  int ranges = 256 / panels;
  if (ranges < 1) ranges = 1;
  const int max_ranges = (M + WS_ROWS - 1) / WS_ROWS;
  if (ranges > max_ranges) ranges = max_ranges;
  const int rows_per = ((M + ranges - 1) / ranges + WS_ROWS - 1) / WS_ROWS * WS_ROWS;
  const int epi = (bias ? WE_BIAS : 0) | (relu ? WE_RELU : 0);
  const dim3 grid(panels * ranges), block(256);
  hipStream_t s = (hipStream_t)stream;
#define WS_LAUNCH(E)                                                                                                         \
  do {                                                                                                                       \
    static bool attr_set = false;                                                                                            \
    if (!attr_set) {                                                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_ws512_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS); \
      if (e != hipSuccess) {                                                                                                 \
        js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));                                                     \
        return JS2T_ERR_LAUNCH;                                                                                              \
      }                                                                                                                      \
      attr_set = true;                                                                                                       \
    }                                                                                                                        \
    hipLaunchKernelGGL(gemm_ws512_kernel<E>, grid, block, WS_LDS, s, (const uint16_t*)X, ldx, (const uint16_t*)W, ldw, (uint16_t*)C, \
                       ldc, bias, M, N, ranges, rows_per);                                                                   \
  } while (0)
  switch (epi) {
    case 0: WS_LAUNCH(0); break;
    case 1: WS_LAUNCH(1); break;
    case 2: WS_LAUNCH(2); break;
    default: WS_LAUNCH(3); break;
  }
#undef WS_LAUNCH
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
