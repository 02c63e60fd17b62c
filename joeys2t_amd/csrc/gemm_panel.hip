// Panel-resident bf16 GEMM for the encoder-sized products with a short reduction (K <= 512):
//   C[M,N] (bf16) = epilogue(A[M,K] B[N,K]^T),  k-contiguous operands  (nn.Linear forward, transformer_layers.py:75-77,147-153,
//   and the input gradient through ReLU + dropout of the feed-forward block's second layer)
//
// Why another kernel.  The persistent 192x128 kernels (gemm.hip) move 40 KB from L2 into LDS per 3.1 MFLOP stage and sit on the
// L2 -> LDS landing rate of a CU (~35 B/clk, MI355X_MICROARCH.md 'gather into LDS'): 1170 cycles of landing against 768 cycles
// of MFMA per stage, measured 1180.  With K <= 512 a whole 96-column panel of B is 96 KB - it fits in LDS next to the A stream,
// so it is fetched ONCE per block instead of once per tile, and what is left to stream is A alone: 4 KB per 0.39 MFLOP
// (10.4 KB / MFLOP against 12.9).  And because B is then read-only, the A stream needs no block-wide ring at all:
//  * a wave owns a strip of 32 rows x the panel's 96 columns (2 x 6 accumulators), streams ITS OWN rows through a private
//    two-slot ring (4 KB per 64-deep stage) by LDS-DMA and orders itself with counted vmcnt waits - there is no barrier in
//    the main loop, the eight waves of a block (two per SIMD) drift apart, and one wave's epilogue (and its wait for the
//    acknowledgement of its stores) runs under the other waves' stages;
//  * a wave's requests run two stages ahead of its multiplies, also across the end of a strip: the stores of a strip's
//    epilogue are YOUNGER than the requests of the next strip's first two stages, so the counted wait for those stages does not
//    wait for the stores (the in-order vmcnt problem of the ring kernels, profiles/r04_epilogue_store_experiments.txt).
//    (Measured and dropped: holding a strip's packed results in registers and storing them INSIDE the next strip's first stage,
//    behind its requests - two and a half stages before any counted wait covers them: no faster, 24 more registers;
//    profiles/r05_pan96_bench.txt.  What the epilogue costs is not the wait for its stores' acknowledgement.)
//  * blocks with equal blockIdx % 8 (one XCD under round-robin placement: speed only) share one eighth of the rows, so A is
//    pulled through the fabric once and every XCD reads all of B (1.5-2 MB).
// Work: unit = (panel, strip).  The blocks of a group cut the group's units, in panel-major order, into equal runs; a block
// loads each panel its run touches (one or two) with two barriers around the load, its wave w takes the run's units
// w, w + 8, ...  LDS: 8 stage images of the panel (96 rows x 128 B, the [row][64 k] image of gemm.hip with the XOR key
// row & 7) = 96 KB + 8 waves x 2 slots x 4 KB = 160 KB: one 512-thread block per CU.
// Column permutation: B fragment j of lane r is panel column 6 r + j: a lane ends up with six consecutive columns of each of its
// rows - ONE 12-byte store per row (sixteen lanes write 192 contiguous bytes), and the dropout decisions of a lane are three whole
// words of the counter hash.  (A first version gave a lane columns 4 r .. 4 r + 3 and 64 + 2 r, 64 + 2 r + 1: two stores per row,
// sixteen per strip - the epilogue's cost is its store instructions, profiles/r05_pan96_variants.txt.)
// Results are bit-identical to the persistent kernels (same k order, same MFMA operand positions, same epilogue arithmetic).
#include "gemm_shared.hpp"

namespace {

typedef int frag_i4 __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t as_bf16x8(const frag_i4& v) { return __builtin_bit_cast(bf16x8_t, v); }

constexpr int PN_COLS = 96, PN_NJ = 6, PN_ROWS = 32, PN_IMG = PN_COLS * 128, PN_NKMAX = 8;
constexpr int PN_BPANEL = PN_NKMAX * PN_IMG;                 // 98304
constexpr int PN_SLOT = PN_ROWS * 128, PN_WRING = 2 * PN_SLOT, PN_WAVES = 8;
constexpr int PN_LDS = PN_BPANEL + PN_WAVES * PN_WRING;      // 163840 = the CU's whole LDS
constexpr int PN_NSTORE = 8;                                 // store instructions of a full strip's epilogue (checked on the ISA)

// LDS-DMA request as inline assembly (common.hpp lds_dma16, plus a memory clobber: the compiler must not move LDS reads across
// it).  Through the builtin the compiler sees a FLAT operation that touches both VMEM and LDS, and while one is pending every wait
// it inserts for one of the kernel's ordinary loads becomes vmcnt(0) - once per strip that would drain the prefetch AND wait for
// the acknowledgement of the epilogue's stores.  Hidden, the requests only make the compiler's counted waits stricter by the
// in-order entries it does not know, never weaker; the kernel orders requests and fragment reads itself.
__device__ __forceinline__ void pan_dma16(const void* gsrc, unsigned char* lds_dst) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(l) : "m0", "memory");
}
__device__ __forceinline__ int pan_col(int j, int r) { return PN_NJ * r + j; }

// FULL: the strip has its 32 rows and the panel its 96 columns - exactly PN_NSTORE store instructions in straight-line code (the
// kernel's counted waits and the compiler's own count both rely on it)
template <int EPI, bool FULL>
__device__ __forceinline__ void pan_store(const js2t_gemm_desc& d, f32x4_t (&acc)[2][PN_NJ], int m0, int n0, int lane,
                                          const float (&bias_r)[PN_NJ], uint32_t drop_key, float my_rs) {
  constexpr bool full = FULL;
#if defined(JS2T_PAN_DBG) && (JS2T_PAN_DBG & 8)  // measurement only: the strip is dropped
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PN_NJ; ++j) asm volatile("" ::"v"(acc[i][j]));
  return;
#endif
  const int g = lane >> 4, r = lane & 15;
  const int M = d.M, N = d.N;
  const int c0 = n0 + PN_NJ * r;               // this lane's six columns c0 .. c0 + 5 (c0 even)
  const bool okc = full || c0 + PN_NJ <= N;     // N % 8 == 0 and c0 % 2 == 0: a lane's columns end inside N or it stores the pairs that do
  constexpr bool has_bias = (EPI & PE_BIAS) != 0, relu = (EPI & PE_RELU) != 0, has_gate = (EPI & PE_GATE) != 0,
                 has_drop = (EPI & PE_DROP) != 0, lnf = (EPI & PE_LNF) != 0;
  const float keep_scale = 1.f / (1.f - d.dropout_p), gate_scale = d.gate_scale;
  const uint32_t thr = (uint32_t)(d.dropout_p * 65536.0f);
  typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
  u32x3_t gq[2][4];
  if (has_gate) {
    const uint16_t* gs = (const uint16_t*)d.gate;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint16_t* gp = gs + (int64_t)min(m0 + 16 * i + 4 * g + e, M - 1) * d.ldg;
        if (full) {
          gq[i][e] = *(const u32x3_t*)(gp + c0);
        } else {  // a lane at the panel's end may own fewer than three column pairs inside N: each pair from its own (clamped) place
#pragma unroll
          for (int k = 0; k < 3; ++k) gq[i][e][k] = *(const uint32_t*)(gp + min(c0 + 2 * k, N - 2));
        }
      }
  }
  uint16_t* Cb = (uint16_t*)d.C;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float ln_rs[4];
    if (lnf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) ln_rs[e] = __shfl(my_rs, 16 * i + 4 * g + e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + 16 * i + 4 * g + e;
      float v[PN_NJ];
#pragma unroll
      for (int j = 0; j < PN_NJ; ++j)
        v[j] = lnf ? fmaf(ln_rs[e], acc[i][j][e], bias_r[j]) : (has_bias ? acc[i][j][e] + bias_r[j] : acc[i][j][e]);
      if (relu) {
#pragma unroll
        for (int j = 0; j < PN_NJ; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      if (has_drop) {  // the decisions of dropout_keep4_key(drop_key, m, c / 4): column c is half c & 1 of hash word c >> 1
        const uint32_t rowkey = hash32((uint32_t)m ^ drop_key) + (uint32_t)(c0 >> 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const uint32_t h = hash32w(rowkey + (uint32_t)q);
          v[2 * q] = (h & 0xffffu) >= thr ? v[2 * q] * keep_scale : 0.f;
          v[2 * q + 1] = (h >> 16) >= thr ? v[2 * q + 1] * keep_scale : 0.f;
        }
      }
      if (has_gate) {
        const u32x3_t q = gq[i][e];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          v[2 * k] = __uint_as_float(q[k] << 16) > 0.f ? v[2 * k] * gate_scale : 0.f;
          v[2 * k + 1] = __uint_as_float(q[k] & 0xffff0000u) > 0.f ? v[2 * k + 1] * gate_scale : 0.f;
        }
      }
      const u32x3_t pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5])};
      uint16_t* cp = Cb + (int64_t)m * d.ldc + c0;
#if defined(JS2T_PAN_DBG) && (JS2T_PAN_DBG & 16)  // measurement only: the epilogue's arithmetic without its stores
      asm volatile("" ::"v"(pk), "v"(cp));
      continue;
#endif
      if (full) {
        *(u32x3_t*)cp = pk;
      } else if (m < M) {
        if (okc) {
          *(u32x3_t*)cp = pk;
        } else {  // the panel's last columns: pairs inside N only
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (c0 + 2 * k < N) *(uint32_t*)(cp + 2 * k) = pk[k];
        }
      }
    }
  }
}

// LayerNorm fold, consumer side (gemm.hip lnf_row_rstd): the eight {sum, sum of squares} pairs of row m0 + (lane & 31) -> that
// row's 1 / sqrt(var + eps), kept in ONE register; the epilogue's lane (g, e) of row block i fetches row 16 i + 4 g + e's value
// from lane 16 i + 4 g + e.  Panel 0 also leaves mean and 1 / sigma for the LayerNorm backward.
__device__ __forceinline__ void lnf_load(const js2t_gemm_desc& d, int m0, int lane, f4_t (&lp)[4]) {
  const f4_t* pp = (const f4_t*)(d.ln_partial + (int64_t)min(m0 + (lane & 31), d.M - 1) * (2 * LNF_GROUPS));
#pragma unroll
  for (int q = 0; q < 4; ++q) lp[q] = pp[q];
}
__device__ __forceinline__ float lnf_finish(const js2t_gemm_desc& d, const f4_t (&lp)[4], int m0, int n0, int lane) {
  const f4_t a = lp[0], b = lp[1], c = lp[2], e = lp[3];
  const float s1 = ((a[0] + a[2]) + (b[0] + b[2])) + ((c[0] + c[2]) + (e[0] + e[2]));
  const float s2 = ((a[1] + a[3]) + (b[1] + b[3])) + ((c[1] + c[3]) + (e[1] + e[3]));
  const float inv = 1.f / (float)(64 * LNF_GROUPS), mu = s1 * inv;
  const float rs = 1.f / sqrtf(fmaxf(fmaf(-mu, mu, s2 * inv), 0.f) + d.ln_eps);
  const int row = m0 + lane;
  if (n0 == 0 && d.ln_mean && lane < PN_ROWS && row < d.M) d.ln_mean[row] = mu, d.ln_rstd[row] = rs;
  return rs;
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pan96_kernel(js2t_gemm_desc d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, g = lane >> 4, r = lane & 15, r8 = lane >> 3, s8 = lane & 7;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int M = d.M, N = d.N, nk = d.K >> 6;  // K % 128 == 0, K <= 512 (launcher)
  const uint16_t* Ab = (const uint16_t*)d.A;
  const uint16_t* Bb = (const uint16_t*)d.B;
  const int64_t lda = d.lda, ldb = d.ldb;
  constexpr bool lnf = (EPI & PE_LNF) != 0, has_bias = (EPI & PE_BIAS) != 0, has_drop = (EPI & PE_DROP) != 0;

  // ---- this block's run of units
  const int G8 = gridDim.x >> 3, xg = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int S = (M + PN_ROWS - 1) / PN_ROWS;
  const int s_lo = (int)(((int64_t)S * xg) >> 3), ns = (int)(((int64_t)S * (xg + 1)) >> 3) - s_lo;
  const int npan = (N + PN_COLS - 1) / PN_COLS;
  const int U = npan * ns;
  const int u0 = (int)((int64_t)U * bi / G8), nu = (int)((int64_t)U * (bi + 1) / G8) - u0;
  if (nu <= 0) return;  // block-uniform, before any barrier

  unsigned char* ring = smem + PN_BPANEL + w * PN_WRING;
  const int lo0 = r * 128 + ((g ^ (r & 7)) << 4), lo1 = r * 128 + (((4 + g) ^ (r & 7)) << 4);

  // ---- request side: two stages ahead of the multiplies
  int ij = w, iks = 0;
  const uint16_t* asrc[4];
  auto set_unit_src = [&](int j) {
    const int u = u0 + j, pan = u / ns;
#if defined(JS2T_PAN_DBG) && (JS2T_PAN_DBG & 1)  // measurement only: every strip reads the rows of the block's first one (L2-resident)
    const int m0 = s_lo * PN_ROWS;
#else
    const int m0 = (s_lo + u - pan * ns) * PN_ROWS;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) asrc[q] = Ab + (int64_t)min(m0 + 8 * q + r8, M - 1) * lda + ((s8 ^ r8) << 3);
  };
  // the four 1 KB pieces of stage f + 2 go out between the MFMA groups of stage f (a request costs its wave 60-100 cycles of issue).
  // Past the end of the stream the requests go on (rows of the last strip into slots nobody reads any more): the loop and its
  // vmcnt counts stay branch-free, for the kernel's waits and for the compiler's
#if defined(JS2T_PAN_DBG) && (JS2T_PAN_DBG & 2)  // measurement only (tools/pan96_variants.sh): no requests, the waves multiply stale LDS
  auto issue_piece = [&](int q) {};
#else
  auto issue_piece = [&](int q) { pan_dma16(asrc[q] + iks * 64, ring + (iks & 1) * PN_SLOT + q * 1024); };
#endif
  auto issue_advance = [&]() {
    if (++iks == nk) {
      iks = 0;
      if (ij + PN_WAVES < nu) ij += PN_WAVES, set_unit_src(ij);
    }
  };
  set_unit_src(min(ij, nu - 1));
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int p = 0; p < 4; ++p) issue_piece(p);
    issue_advance();
  }

  const uint32_t drop_key = has_drop ? dropout_key(d.rng_state, d.rng_stream) : 0u;
  f32x4_t acc[2][PN_NJ];
  frag_i4 ax[2][2], ay[2][2], bx[PN_NJ], by[PN_NJ];
  float bias_r[PN_NJ];
  bool a_ready = false;        // ax holds the fragments of stage f
  bool stores_pending = false; // a full strip's PN_NSTORE stores were issued after the requests of stages f and f + 1
  float my_rs = 0.f;

  auto read_a = [&](int slot, frag_i4 (&a)[2][2]) {
    const unsigned char* sp = ring + slot * PN_SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      a[i][0] = *(const frag_i4*)(sp + i * 2048 + lo0);
      a[i][1] = *(const frag_i4*)(sp + i * 2048 + lo1);
    }
  };

  const int p_first = u0 / ns, p_last = (u0 + nu - 1) / ns;
  for (int pan = p_first; pan <= p_last; ++pan) {
    const int ja = max(pan * ns, u0) - u0, jb = min((pan + 1) * ns, u0 + nu) - u0;
    const int n0 = pan * PN_COLS;
    // ---- the panel: every wave is through with the previous one; wave w fetches stage image w
    __syncthreads();
    if (w < nk) {
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int j = q >> 1, rr = (q & 1) * 8 + r8;
        const uint16_t* sp = Bb + (int64_t)min(n0 + pan_col(j, rr), N - 1) * ldb + w * 64 + ((s8 ^ r8) << 3);
        pan_dma16(sp, smem + w * PN_IMG + q * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < PN_NJ; ++j) {
      const int c = n0 + pan_col(j, r);
      bias_r[j] = has_bias ? d.bias[min(c, N - 1)] : 0.f;  // columns >= N are never stored
    }
    const int j_first = ja + ((w - ja) & (PN_WAVES - 1));  // this wave's first unit of the run's part in this panel
    f4_t lp0[4];
    if (lnf && j_first < jb) lnf_load(d, (s_lo + u0 + j_first - pan * ns) * PN_ROWS, lane, lp0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the panel, and everything this wave had in flight
    __syncthreads();
    if (lnf && j_first < jb) my_rs = lnf_finish(d, lp0, (s_lo + u0 + j_first - pan * ns) * PN_ROWS, n0, lane);
    // the compiler does not see the wait above: left at that, it would wait for the bias registers in every epilogue, behind
    // whatever that epilogue has just requested.  A use here puts its own (satisfied) wait here.
#pragma unroll
    for (int j = 0; j < PN_NJ; ++j) asm volatile("" : "+v"(bias_r[j]));
    stores_pending = false;
    bool b_stale = true;  // bx belongs to the previous panel

    for (int j = j_first; j < jb; j += PN_WAVES) {
      const int u = u0 + j, m0 = (s_lo + u - pan * ns) * PN_ROWS;
      const bool full = m0 + PN_ROWS <= M && n0 + PN_COLS <= N;
      if (!a_ready) {  // first stage of the stream: it landed before the vmcnt(0) above
        read_a(0, ax);
        a_ready = true;
      }
      if (b_stale) {
#pragma unroll
        for (int jj = 0; jj < PN_NJ; ++jj) bx[jj] = *(const frag_i4*)(smem + jj * 2048 + lo0);
        b_stale = false;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < PN_NJ; ++jj) acc[i][jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};

      auto stage = [&](int ks, frag_i4 (&ac)[2][2], frag_i4 (&an)[2][2]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ac and bx are in registers: the slot of this stage is free
        const unsigned char* bimg = smem + ks * PN_IMG;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#if !(defined(JS2T_PAN_DBG) && (JS2T_PAN_DBG & 4))  // & 4, measurement only: MFMAs on stale B fragments
          by[2 * q] = *(const frag_i4*)(bimg + (2 * q) * 2048 + lo1);
          by[2 * q + 1] = *(const frag_i4*)(bimg + (2 * q + 1) * 2048 + lo1);
#endif
          if (q < 2) issue_piece(2 * q), issue_piece(2 * q + 1);  // the stage after the next one, into this stage's slot
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 2 * q; jj < 2 * q + 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ac[i][0]), as_bf16x8(bx[jj]), acc[i][jj], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        issue_advance();
        // the next stage has landed: everything older than the four requests just made is done - after a full strip's epilogue
        // the stores (and the next strip's row statistics, requested in front of them) are younger than the awaited requests too
        if (ks == 0 && stores_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + PN_NSTORE + (lnf ? 4 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        read_a((ks + 1) & 1, an);
        const unsigned char* bnext = smem + (ks + 1 == nk ? 0 : ks + 1) * PN_IMG;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#if !(defined(JS2T_PAN_DBG) && (JS2T_PAN_DBG & 4))
          bx[2 * q] = *(const frag_i4*)(bnext + (2 * q) * 2048 + lo0);
          bx[2 * q + 1] = *(const frag_i4*)(bnext + (2 * q + 1) * 2048 + lo0);
#endif
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 2 * q; jj < 2 * q + 2; ++jj)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ac[i][1]), as_bf16x8(by[jj]), acc[i][jj], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      for (int ks = 0; ks < nk; ks += 2) {
        stage(ks, ax, ay);
        stage(ks + 1, ay, ax);
      }
      stores_pending = false;
      // epilogue; the NEXT strip's row statistics are requested in front of the stores and turned into 1 / sigma behind them (a load
      // consumed stages later would make the compiler wait for the requests in between)
      const bool more = j + PN_WAVES < jb;
      const int m0n = m0 + PN_WAVES * PN_ROWS;  // same panel: the strips of a run are consecutive
      if (full) {
        f4_t lp[4];
        if (lnf && more) lnf_load(d, m0n, lane, lp);
        pan_store<EPI, true>(d, acc, m0, n0, lane, bias_r, drop_key, my_rs);
        __builtin_amdgcn_sched_barrier(0);
        if (lnf && more) my_rs = lnf_finish(d, lp, m0n, n0, lane);
        stores_pending = more;
      } else {
        pan_store<EPI, false>(d, acc, m0, n0, lane, bias_r, drop_key, my_rs);
        if (lnf && more) {
          f4_t lp[4];
          lnf_load(d, m0n, lane, lp);
          my_rs = lnf_finish(d, lp, m0n, n0, lane);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#define g_pan_mode js2t_ctx_value(JS2T_CTX_GEMM_PANEL_MODE)  // js2t_gemm_panel_mode / JS2T_CTX_GEMM_PANEL_MODE: -1 = products big enough for it, 0 = never, 1 = every product that qualifies

template <int EPI>
int launch_pan96_epi(const js2t_gemm_desc& d, hipStream_t s) {
  static int n_cu = 0;
  if (n_cu == 0) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_pan96_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, PN_LDS);
    int dev = 0, cu = 0;
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu < 8) {
      js2t_set_error("gemm pan96 setup: %s", hipGetErrorString(e));
      return JS2T_ERR_LAUNCH;
    }
    n_cu = cu & ~7;
  }
  hipLaunchKernelGGL((gemm_bf16_pan96_kernel<EPI>), dim3(n_cu), dim3(512), PN_LDS, s, d);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

inline bool al(const void* p, uintptr_t a) { return (((uintptr_t)p) & (a - 1)) == 0; }

}  // namespace

extern "C" void js2t_gemm_panel_mode(int mode) { js2t_ctx_override(JS2T_CTX_GEMM_PANEL_MODE, mode < 0 ? -1 : (mode > 0 ? 1 : 0)); }

int launch_bf16_pan96(const js2t_gemm_desc& d, int mask, hipStream_t s) {
  if (g_pan_mode == 0) return -1;
  if (d.dtype_ab != JS2T_BF16 || d.dtype_c != JS2T_BF16 || d.trans_a || d.trans_b || d.conv || d.split_k > 1 || d.batch != 1) return -1;
  if (d.alpha != 1.f || d.alpha_dev || d.preact || d.beta != 0.f || d.a_rowsum || d.residual || d.rs_partial || d.dot_partial || d.c8) return -1;
  if ((d.K & 127) || d.K < 128 || d.K > 64 * PN_NKMAX || (d.N & 7) || d.N < PN_COLS || d.M < PN_ROWS) return -1;
  if (!al(d.A, 16) || !al(d.B, 16) || (d.lda & 7) || (d.ldb & 7) || !al(d.C, 8) || (d.ldc & 3)) return -1;
  if (d.gate && (!al(d.gate, 8) || (d.ldg & 3))) return -1;
  if (d.ln_partial && (d.K != 64 * LNF_GROUPS || !al(d.ln_partial, 16))) return -1;
  // worth it from two units per wave on (a block loads 96 KB of B before its first multiply)
  const int64_t units = (int64_t)((d.M + PN_ROWS - 1) / PN_ROWS) * ((d.N + PN_COLS - 1) / PN_COLS);
  if (g_pan_mode < 0 && units < 16 * 256) return -1;
  // ... and not with dropout in the epilogue: the strip's hash arithmetic (13 vector instructions per element) sits in the same wave
  // as its multiplies and the partner wave does not cover it - FFN layer 1 takes 36-37 us here against 32-33 on the persistent kernel
  // (profiles/r05_pan96_bench.txt); the projections and the gated input gradient gain 4-15 %
  if (g_pan_mode < 0 && (mask & PE_DROP)) return -1;
  switch (mask) {
    case 0: return launch_pan96_epi<0>(d, s);
    case PE_BIAS: return launch_pan96_epi<PE_BIAS>(d, s);
    case PE_BIAS | PE_LNF: return launch_pan96_epi<PE_BIAS | PE_LNF>(d, s);
    case PE_BIAS | PE_RELU | PE_DROP | PE_LNF: return launch_pan96_epi<PE_BIAS | PE_RELU | PE_DROP | PE_LNF>(d, s);
    case PE_BIAS | PE_RELU | PE_LNF: return launch_pan96_epi<PE_BIAS | PE_RELU | PE_LNF>(d, s);
    case PE_BIAS | PE_RELU | PE_DROP: return launch_pan96_epi<PE_BIAS | PE_RELU | PE_DROP>(d, s);
    case PE_GATE: return launch_pan96_epi<PE_GATE>(d, s);
    default: return -1;
  }
}
