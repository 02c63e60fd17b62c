// Fused ("flash") multi-head attention for bf16, head sizes 128 and 64 — forward, dK/dV and dQ kernels.
//
// Replaces the materialised chain of MultiHeadedAttention.forward (transformer_layers.py:86-105):
//   scores = (q/sqrt(dh)) k^T ; masked_fill(~mask, -inf) ; softmax ; dropout ; @ v
// and its autograd backward.  Scores and probabilities never touch HBM (the [B,H,T,T] tensors were ~5 passes of
// 72 MB per layer); K/V (forward, dQ) or Q/dO (dK/dV) tiles are staged in LDS by LDS-DMA, QK^T / PV / the three
// backward products run on v_mfma_f32_16x16x32_bf16, the online softmax reduces with wavefront shuffles.
//
// Geometry (all three kernels): 256 threads = 4 waves; a wave owns 32 "own" rows (two 16-row MFMA column blocks) of
// one (batch, head) and sweeps the other sequence in 64-row tiles held in LDS as [64][128] bf16 images (256 B rows,
// 16-byte chunk c of row r stored at slot c ^ ((r&7)<<1): conflict-free for both ds_read_b128 row reads and
// ds_read_b64_tr_b16 transposed reads, applied on the DMA source address).  MFMAs are issued "swapped"
// (D rows = tile rows / head columns, D cols = own rows) so that each lane holds, for ONE own row, 4 consecutive
// tile rows — the probability registers are then directly the B operand of the next product, with the k index
// permuted as {sub-tile 2s rows 4g..4g+3, sub-tile 2s+1 rows 4g..4g+3}; the transposed LDS reads use the same order.
//
// Head size 64 (mustc_st.yaml: 8 heads of 64) uses the SAME 16 KiB image and the same access patterns: a tile is then 128
// rows of the swept sequence, rows 0-63 in chunk columns 0-7 and rows 64-127 in chunk columns 8-15 of the [64][128]
// image (the k-steps over the head columns halve, the tile rows double: 32 MFMAs per wave and tile either way).
//
// Dropout uses the same counter hash and (row, col4) indexing as softmax_fwd_kernel, so fused and unfused paths
// draw identical masks (tests compare them).  A row whose keys are all masked yields NaN like the reference.
#include "common.hpp"

namespace {

constexpr int IMG_BYTES = 64 * 128 * 2;  // 16 KiB
// geometry of one head size: KT tile rows (of the swept sequence) per image, NTT 16-row blocks, NKS 32-wide k-steps over
// the head columns, NCT 16-column blocks of the head, NSS 32-row k-steps over the tile rows, CPR 16-byte chunks per row
template <int DH> struct Geo {
  static_assert(DH == 128 || DH == 64, "head size");
  static constexpr int KT = 8192 / DH, NTT = KT / 16, NKS = DH / 32, NCT = DH / 16, NSS = KT / 32, CPR = DH / 8;
  static constexpr uint32_t FULL = NTT == 8 ? 0xffffffffu : 0xffffu;
};
typedef __attribute__((address_space(1))) const void g_cvoid;
typedef __attribute__((address_space(3))) void l_void;
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

struct AttnArgs {
  const uint16_t* q; const uint16_t* k; const uint16_t* v; const uint16_t* o; const uint16_t* d_o;
  uint16_t* out;   // fwd: O ; bwd: unused
  uint16_t* dq; uint16_t* dk; uint16_t* dv;
  float* lse;      // [B*H, Tq]
  float* delta;    // [B*H, Tq] rowsum(dO * O)
  const uint8_t* mask;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, msb, msq;
  int B, H, Tq, Tk;
  float scale, p;
  const uint64_t* rng;
  uint32_t stream;
  const float* rel;  // relative-position bias f32[H, 2R+1] (NULL: none)
  float* d_rel;
  int relR;
  const float* dpart;  // bwd: delta as partial sums over 64-column groups of [B*Tq, H*DH] (js2t_attn_desc.delta_partial) or NULL
  int dgroups;
  const int32_t* seg;  // packed rows (js2t_attn_desc.seg): entry b owns rows seg[b] .. seg[b+1] of every buffer; NULL: b * Tq ..
  unsigned long long* d_rel_fix;  // deterministic mode: the bias gradient as 2^-32 fixed-point sums [H, 2R+1] (common.hpp), else NULL
  int seg_rows;  // packed rows: rows of the buffers; seg[B] .. seg_rows belong to nobody and are zeroed (0: left alone)
  int seg_keys;  // seg describes the key side only (cross-attention over packed keys): queries stay padded
};
// where batch entry b lives: first row in the query-side and key-side buffers, its own lengths.  a.Tq / a.Tk stay the PADDED
// lengths: grid shape, [B*H, Tq] scalars (lse, delta), mask rows and the dropout counter (z * Tq + q: the packed layout draws the
// masks of the padded one) are indexed with them
struct Seg { int q0, k0, Tq, Tk; };
__device__ __forceinline__ Seg seg_of(const AttnArgs& a, int b) {
  if (a.seg) {
    const int r0 = a.seg[b], n = a.seg[b + 1] - r0;
    if (a.seg_keys) return Seg{b * a.Tq, r0, a.Tq, n};
    return Seg{r0, r0, n, n};
  }
  return Seg{b * a.Tq, b * a.Tk, a.Tq, a.Tk};
}
// delta = rowsum(dO * O) of query row qc of head (b, h): from the partial sums the product that made dO left behind, or the
// [B*H, Tq] array the dQ pass writes
template <int DH>
__device__ __forceinline__ float load_delta(const AttnArgs& a, int row0, int h, int z, int qc) {
  if (a.dpart) {
    const float* pp = a.dpart + ((int64_t)row0 + qc) * a.dgroups + h * (DH / 64);
    return DH == 128 ? pp[0] + pp[1] : pp[0];
  }
  return a.delta[(int64_t)z * a.Tq + qc];
}
// packed rows: the rows behind the last entry (js2t_attn_desc.seg_rows), zeroed by the grid as a whole - block bid of nblk takes
// rows seg[B] + bid, + nblk, ..: at most one row each for an encoder layer's grid (a launch of its own cost 4.8 us, 32 times a step)
__device__ __forceinline__ void zero_tail_rows(const AttnArgs& a, uint16_t* base, int64_t ld, int width, int bid, int nblk) {
  for (int row = a.seg[a.B] + bid; row < a.seg_rows; row += nblk)
    for (int c = 4 * (int)threadIdx.x; c < width; c += 1024) *(uint2*)(base + (int64_t)row * ld + c) = make_uint2(0u, 0u);
}
constexpr int REL_MAX = 255;  // largest clipping distance: the per-head table (2 * 255 + 1 floats) is staged in LDS
// the head's bias table in base-2 units, staged once per block
__device__ __forceinline__ void stage_rel(float* rel_s, const AttnArgs& a, int h, int t) {
  const int n = 2 * a.relR + 1;
  for (int i = t; i < n; i += 256) rel_s[i] = a.rel[(int64_t)h * n + i] * 1.4426950408889634f;
}
__device__ __forceinline__ int rel_index(int key, int query, int R) { return min(max(key - query, -R), R) + R; }

// HBM -> LDS image of KT rows starting at row0 (rows clamped to rows_max-1), 4 DMA pieces per wave.
// The requests are hidden from the compiler (lds_dma16, common.hpp): with the builtin in a block every wait the compiler places
// for its LDS reads becomes lgkmcnt(0) - a full drain of the fragment reads in flight in front of every other product - and every
// transposing read gets a vmcnt(0).  The kernels order requests and reads themselves: vmcnt(0) + barrier where an image is first read.
// What does not change from tile to tile is computed ONCE per block (ImgSrc): this lane's tile row and column in each of its
// wave's four pieces.  Per tile and piece that leaves one add for a whole tile, a clamp more for the last one - the first
// version rebuilt a 64-bit address per piece and tile (two 32 x 32 multiplies, a 64-bit multiply-add, shifts: ten VALU
// instructions x 8-16 pieces per tile, a fifth of the forward kernel's vector instructions).  ld < 2^17 (check_common) and at
// most 8192 rows per entry keep the element offset inside 31 bits; the entry's base pointer is wave-uniform.
template <int DH> struct ImgSrc {
  int trow[4];  // tile row of piece q's 16 bytes
  int col[2];   // first head column, pieces q & 1
};
template <int DH>
__device__ __forceinline__ ImgSrc<DH> img_src(int w, int lane) {
  ImgSrc<DH> s;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = (w * 4 + q) * 64 + lane;
    const int row = p >> 4, slot = p & 15;
    const int chunk = slot ^ ((row & 7) << 1);
    s.trow[q] = DH == 128 ? row : row + 64 * (chunk >> 3);
    if (q < 2) s.col[q] = DH == 128 ? chunk * 8 : (chunk & 7) * 8;
  }
  return s;
}
template <int DH>
__device__ __forceinline__ void img_dma(const uint16_t* base, int ld, int row0, int rows_max, unsigned char* img, int w,
                                        const ImgSrc<DH>& is) {
  if (row0 + Geo<DH>::KT <= rows_max) {  // wave-uniform
    const int o0 = row0 * ld;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t off = (uint32_t)(o0 + is.trow[q] * ld + is.col[q & 1]);
      lds_dma16(base + off, img + (w * 4 + q) * 1024);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t off = (uint32_t)(min(row0 + is.trow[q], rows_max - 1) * ld + is.col[q & 1]);
      lds_dma16(base + off, img + (w * 4 + q) * 1024);
    }
  }
}
// ONE of a wave's four pieces, as a request the compiler does not see (lds_dma16, common.hpp): for kernels that spread the next
// tile's requests over the current tile's products - a burst of eight behind the barrier holds its wave for ~1100 ticks while the
// CU's one address path takes the requests of eight waves in turn (profiles/r05_attn_bwd_ticks.txt), and the builtin form makes the
// compiler drain every request in front of the first transposing read.  The caller orders requests and reads itself (counted
// vmcnt + barrier at the top of its loop).
template <int DH>
__device__ __forceinline__ void img_dma_piece(const uint16_t* base, int ld, int row0, int rows_max, unsigned char* img, int w,
                                              const ImgSrc<DH>& is, int q) {
  const int row = min(row0 + is.trow[q], rows_max - 1);  // (no fast path: a branch here cuts the caller's product loop into blocks)
  lds_dma16(base + (uint32_t)(row * ld + is.col[q & 1]), img + (w * 4 + q) * 1024);
}
// row fragment: 8 consecutive head columns (32*ks + 8*(lane>>4) ..) of tile row 16*tt + (lane&15)
template <int DH>
__device__ __forceinline__ bf16x8_t img_row(const unsigned char* img, int tt, int ks, int lane) {
  const int row = ((16 * tt) & 63) + (lane & 15), chunk = (tt >> 2) * Geo<DH>::CPR + 4 * ks + (lane >> 4);
  return *(const bf16x8_t*)(img + row * 256 + ((chunk ^ ((row & 7) << 1)) << 4));
}
// transposed fragment for k-step s and head-column tile ct: element j<4 = row 32s+4g+j, j>=4 = row 32s+16+4g+(j-4),
// column 16*ct + (lane&15)
template <int DH>
__device__ __forceinline__ bf16x8_t img_tr(const unsigned char* img, int s, int ct, int lane) {
  const int gl = lane & 15, qq = gl >> 2, p = gl & 3, g = lane >> 4;
  const int r1 = ((32 * s) & 63) + 4 * g + qq, r2 = r1 + 16;
  const int cc = (s >> 1) * Geo<DH>::CPR + 2 * ct + (p >> 1), off = 8 * (p & 1);
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + r1 * 256 + ((cc ^ ((r1 & 7) << 1)) << 4) + off));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + r2 * 256 + ((cc ^ ((r2 & 7) << 1)) << 4) + off));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}
// own-row fragments straight from HBM: rows rb + (lane&15) (clamped), columns 32*ks + 8*(lane>>4) ..
template <int NKS>
__device__ __forceinline__ void own_frags(const uint16_t* base, int64_t ld, int rb, int rows_max, int lane, bf16x8_t (&f)[NKS]) {
  const int row = min(rb + (lane & 15), rows_max - 1);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) f[ks] = *(const bf16x8_t*)(base + (int64_t)row * ld + 32 * ks + 8 * (lane >> 4));
}
// The compiler waits for a global load where its result is first used.  With the LDS-DMA requests hidden from it (img_dma) such a
// wait inside a tile loop - vmcnt(N) for "its" loads - would also wait for the younger requests of the tile in flight: the own-row
// fragments are therefore "used" once in front of the loop, which moves their wait there.
template <int N>
__device__ __forceinline__ void pin_loaded(bf16x8_t (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i]));
}
__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& a, const f32x4_t& b) {
  const uint4 v = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
  return __builtin_bit_cast(bf16x8_t, v);
}
// Reductions over the 4 lanes that share (lane & 15), i.e. over the wave's four 16-lane rows.  gfx950's row swaps do the
// exchange in the VALU: v_permlane16_swap(a, b) trades the odd rows of a for the even rows of b, v_permlane32_swap the upper
// half of a for the lower half of b - applied to two copies of v, the pair holds {own, partner} in every lane.
// (__shfl_xor compiles to ds_bpermute_b32 + s_waitcnt lgkmcnt(0): two LDS round trips per reduction that also drain the
// fragment reads in flight.)
__device__ __forceinline__ void row_partners16(float v, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void row_partners32(float v, float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float quad_max(float v) {
  float a, b;
  row_partners16(v, a, b);
  v = fmaxf(a, b);
  row_partners32(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float quad_sum(float v) {
  float a, b;
  row_partners16(v, a, b);
  v = a + b;
  row_partners32(v, a, b);
  return a + b;
}
__device__ __forceinline__ void store4(uint16_t* p, const f32x4_t& v, float sc) {
  uint2 pk;
  pk.x = pack_bf16x2(v[0] * sc, v[1] * sc);
  pk.y = pack_bf16x2(v[2] * sc, v[3] * sc);
  *(uint2*)p = pk;
}

// Key validity for key-padding masks (mask_sq == 0) is staged ONCE per block in LDS, already in the form the tiles want it: one
// 16-bit word per (64-key chunk c, lane group g) with bit 4 tt + r <-> key 64 c + 16 tt + 4 g + r, the words of a tile side by
// side - per tile a lane then reads ONE word (the byte-per-key form cost four LDS reads and ~30 bit operations per tile).  A wave
// takes 64 consecutive keys; its ballot is the chunk, lanes 0-3 cut the four words out of it.
constexpr int KMASK_MAX = 8192;
template <int DH>
__device__ __forceinline__ void stage_kbits(uint16_t* kb, const AttnArgs& a, int b, int Tk, int nkeys_padded, int t) {
  constexpr int HV = Geo<DH>::KT / 64;  // 64-key chunks per tile
  const int lane = t & 63;
  for (int k0 = t & ~63; k0 < nkeys_padded; k0 += 256) {
    const int k = k0 + lane;
    bool on = k < Tk;
    if (on && a.mask && a.msq == 0) on = a.mask[(int64_t)b * a.msb + k] != 0;
    const uint64_t bal = __ballot(on);
    if (lane < 4) {
      const uint64_t v = bal >> (4 * lane);
      const uint32_t w16 = (uint32_t)(v & 15u) | ((uint32_t)(v >> 12) & 0xf0u) | ((uint32_t)(v >> 24) & 0xf00u) | ((uint32_t)(v >> 36) & 0xf000u);
      const int c = k0 >> 6;
      kb[((c / HV) * 4 + lane) * HV + (c % HV)] = (uint16_t)w16;
    }
  }
}
// validity bits of this lane's keys in tile kt: bit (4*tt + r) <-> key KT*kt + 16tt + 4g + r
template <int DH>
__device__ __forceinline__ uint32_t tile_kbits(const uint16_t* kb, int kt, int g) {
  if (DH == 128) return kb[kt * 4 + g];
  return *(const uint32_t*)(kb + (kt * 4 + g) * 2);
}
// per-query full mask (mask_sq != 0): AND the row's bytes into the key bits
template <int DH>
__device__ __forceinline__ uint32_t row_kbits(uint32_t bits, const uint8_t* mrow, int kt, int g, int Tk) {
  // all bytes are requested before the first is looked at (clamped addresses, no branch): written as
  // `if (key < Tk && !mrow[key])` this compiled to 16-32 byte loads each waited for in turn
  uint8_t mv[4 * Geo<DH>::NTT];
#pragma unroll
  for (int tt = 0; tt < Geo<DH>::NTT; ++tt)
#pragma unroll
    for (int r = 0; r < 4; ++r) mv[4 * tt + r] = mrow[min(Geo<DH>::KT * kt + 16 * tt + 4 * g + r, Tk - 1)];
#pragma unroll
  for (int i = 0; i < 4 * Geo<DH>::NTT; ++i) bits &= mv[i] ? 0xffffffffu : ~(1u << i);
  return bits;
}

#ifdef JS2T_ATTN_PROF  // s_memtime probes of one wave (tools/attn_fwd_prof.py); not part of the normal build
__device__ unsigned long long g_attn_prof[8];
#define ATT_T(i) do { const unsigned long long c_ = __builtin_readcyclecounter(); prof_[i] += c_ - last_; last_ = c_; } while (0)
#define ATT_PIN(v) asm volatile("s_nop 0" ::"v"(v))
#else
#define ATT_T(i)
#define ATT_PIN(v)
#endif
// ------------------------------------------------------------------------------------------------ forward
// a wave owns 16 queries; a block 64
// SB (single-buffered): ONE K image and ONE V image per block (32 KB) and the key mask in dynamic LDS, so that three
// blocks fit a CU (three waves per SIMD at <= 168 registers) and the 768 blocks of an encoder layer are resident at once.
// V(kt) is requested when K(kt) has landed and lands under QK^T + softmax; K(kt+1) is requested when every wave is through
// with QK^T and lands under PV (what of it is exposed, the other two blocks of the CU cover): two barriers per tile.
// DROP: dropout compiled in or out - as a run-time flag every 16-key block of the element loops had its own branch around the hashes,
// i.e. its own basic block with an LDS wait at its head, and nothing of one block's arithmetic could hide under another's products
template <int DH, bool REL, bool SB = false, bool DROP = true>
__global__ __launch_bounds__(256, SB ? 3 : 2) void flash_fwd_kernel(AttnArgs a) {
  using G = Geo<DH>;
  constexpr int KT = G::KT, NTT = G::NTT, NKS = G::NKS, NCT = G::NCT, NSS = G::NSS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages x {K image, V image}; SB: K, V, key mask
  __shared__ __attribute__((aligned(16))) uint16_t kmask_st[SB ? 8 : KMASK_MAX / 16];
  uint16_t* kmask = SB ? (uint16_t*)(smem + 2 * IMG_BYTES) : kmask_st;
  __shared__ float rel_s[REL ? 2 * REL_MAX + 1 : 1];
#ifdef JS2T_ATTN_PROF
  const unsigned long long t_start_ = __builtin_readcyclecounter();
#endif
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4, m = lane & 15;
  const int wu = __builtin_amdgcn_readfirstlane(w);  // wave-uniform copy: LDS-DMA destinations stay on the scalar unit
  const ImgSrc<DH> isrc = img_src<DH>(wu, lane);
  // 1-D grid, XCD-aware: the query tiles of one (batch, head) pair are neighbours in the logical order, so they run on
  // one XCD and its L2 serves their common K / V (round-robin placement had every XCD fetch every head's K / V: six times
  // the fabric traffic, which is what bound these kernels)
  const int ntile = (a.Tq + 63) / 64;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = lid / ntile, b = z / a.H, h = z - b * a.H;
  const int q0 = (lid - z * ntile) * 64 + w * 16;
  const Seg sg = seg_of(a, b);
  const int Tq = sg.Tq, Tk = sg.Tk;
  if (a.seg_rows && !a.seg_keys) zero_tail_rows(a, a.out, a.ldo, a.H * DH, blockIdx.x, gridDim.x);
  if (q0 - w * 16 >= Tq) return;  // packed rows: a tile behind this utterance's last query (block-uniform, before any barrier)
  const uint16_t* Qb = a.q + (int64_t)sg.q0 * a.ldq + h * DH;
  const uint16_t* Kb = a.k + (int64_t)sg.k0 * a.ldk + h * DH;
  const uint16_t* Vb = a.v + (int64_t)sg.k0 * a.ldv + h * DH;
  // the first K (V) image is requested before anything else is fetched: the key mask below waits for its own loads before it can
  // write them to LDS, and a request issued behind that wait starts a second memory round trip where one would do
  img_dma<DH>(Kb, (int)a.ldk, 0, Tk, smem, wu, isrc);
  if (!SB) img_dma<DH>(Vb, (int)a.ldv, 0, Tk, smem + IMG_BYTES, wu, isrc);
  bf16x8_t qf[NKS];
  own_frags<NKS>(Qb, a.ldq, q0, Tq, lane, qf);
  f32x4_t o[NCT];
  float mi = -INFINITY, li = 0.f;
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr bool drop = DROP;  // (launched with DROP = dropout_p > 0)
  const uint32_t dkey = drop ? dropout_key(a.rng, a.stream) : 0u;
  const float drop_sc = drop ? 1.f / (1.f - a.p) : 1.f;
  const float scale2 = a.scale * 1.4426950408889634f;
  const int nkt = (Tk + KT - 1) / KT;
  const bool full_mask = a.mask && a.msq != 0;
  // the dropout word of (row, col4 c4, half w) is hash32w(rowkey + 2 c4 + w): kept as (..) * M1 and advanced by adds (hash32w_pre)
  const uint32_t rowkeym = (hash32((uint32_t)(z * a.Tq + min(q0 + m, Tq - 1)) ^ dkey) + 2u * (uint32_t)g) * HASH32W_M1;
  const uint32_t thr = (uint32_t)(a.p * 65536.0f);
  stage_kbits<DH>(kmask, a, b, Tk, nkt * KT, t);
  if (REL) stage_rel(rel_s, a, h, t);
  pin_loaded(qf);
  int cur = 0;
#ifdef JS2T_ATTN_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = t_start_;
  ATT_T(6);  // prologue: own query fragments, key mask, first K / V request
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_T(0);
    __syncthreads();
    ATT_T(1);
    if (SB) {
      img_dma<DH>(Vb, (int)a.ldv, kt * KT, Tk, smem + IMG_BYTES, wu, isrc);  // every wave is through with PV of the previous tile
    } else if (kt + 1 < nkt) {
      img_dma<DH>(Kb, (int)a.ldk, (kt + 1) * KT, Tk, smem + (cur ^ 1) * 2 * IMG_BYTES, wu, isrc);
      img_dma<DH>(Vb, (int)a.ldv, (kt + 1) * KT, Tk, smem + (cur ^ 1) * 2 * IMG_BYTES + IMG_BYTES, wu, isrc);
    }
    ATT_T(2);
    const unsigned char* Ki = SB ? smem : smem + cur * 2 * IMG_BYTES;
    const unsigned char* Vi = Ki + IMG_BYTES;
    const uint32_t kbits = tile_kbits<DH>(kmask, kt, g);
    // S^T = K Q^T : s[tt][r] = score(query q0+m, key KT*kt + 16tt + 4g + r)
    f32x4_t s[NTT];
#pragma unroll
    for (int tt = 0; tt < NTT; ++tt) s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    {  // every product takes a fresh K fragment from LDS: the reads run two fragments ahead of the products (pinned by scheduling
       // groups; left alone the compiler drained the LDS queue - lgkmcnt(0) - in front of every other product)
      constexpr int NF = NTT * NKS;
      bf16x8_t fk[3];
      fk[0] = img_row<DH>(Ki, 0, 0, lane);
      fk[1] = img_row<DH>(Ki, 1 / NKS, 1 % NKS, lane);
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        if (i + 2 < NF) fk[(i + 2) % 3] = img_row<DH>(Ki, (i + 2) / NKS, (i + 2) % NKS, lane);
        s[i / NKS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[i % 3], qf[i % NKS], s[i / NKS], 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int i = 0; i < NF - 2; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
    // online softmax in base 2: mi = running max of s*scale*log2e, li = running sum of exp2(.. - mi); the dropped
    // probabilities go to the PV product unscaled, 1/(1-p) is applied with 1/li at the end
    const bool tile_clear = !full_mask && __all(kbits == G::FULL) != 0;  // wave-uniform: every key of the tile is live
    ATT_PIN(s[NTT - 1][3]);
    ATT_T(3);
    bf16x8_t pf[NSS];
    {
      uint32_t bits = kbits;
      if (REL) {  // scores go to base-2 units here, the bias of (query, key) with them: the maximum is taken over the sum
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[tt][r] = fmaf(s[tt][r], scale2, rel_s[rel_index(KT * kt + 16 * tt + 4 * g + r, q0 + m, a.relR)]);
      }
      if (!tile_clear) {
        const int qc = min(q0 + m, Tq - 1);
        if (full_mask) bits = row_kbits<DH>(bits, a.mask + (int64_t)b * a.msb + (int64_t)qc * a.msq, kt, g, Tk);
#pragma unroll
        for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[tt][r] = ((bits >> (4 * tt + r)) & 1u) ? s[tt][r] : -INFINITY;
      }
#ifdef JS2T_ATTN_NOMAX
      // measurement build (profiles/README.md, "what a one-pass softmax could save"): no running maximum, no rescale --
      // the cost floor of any scheme that knows the row maximum before the key loop starts. Right only for bounded scores.
      const float mnew = 0.f, msafe = 0.f, corr = 1.f;
#else
      float mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[tt][r]);
      mx = quad_max(mx) * (REL ? 1.f : scale2);  // scale > 0: the max commutes with it
      const float mnew = fmaxf(mi, mx);
      const float msafe = mnew == -INFINITY ? 0.f : mnew;
      const float corr = __builtin_amdgcn_exp2f(mi - msafe);  // exp2(-inf) = 0 on the first live tile
#endif
      float rs = 0.f;
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu;
        if (drop) {  // == dropout_keep4_key(dkey, z*Tq + q, col4 = (KT/4) kt + 4 tt + g) with the row hash hoisted out of the key loop
          const uint32_t xm = rowkeym + (uint32_t)kt * ((KT / 2) * HASH32W_M1) + (uint32_t)tt * (8u * HASH32W_M1);
          h0 = hash32w_pre(xm), h1 = hash32w_pre(xm + HASH32W_M1);
        }
        const uint32_t hv[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(REL ? s[tt][r] - msafe : fmaf(s[tt][r], scale2, -msafe));
          rs += pv;
          s[tt][r] = hv[r] >= thr ? pv : 0.f;
        }
      }
      rs = quad_sum(rs);
      li = li * corr + rs;
      mi = mnew;
      if (!__all(corr == 1.f)) {  // the running maximum moved for some row of this wave: rescale the accumulators
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[ct][r] *= corr;
      }
#pragma unroll
      for (int ss = 0; ss < NSS; ++ss) pf[ss] = pack8(s[2 * ss], s[2 * ss + 1]);
    }
    ATT_PIN(pf[NSS - 1]);
    ATT_T(4);
    if (SB) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of V(kt)
      __syncthreads();                                   // V(kt) complete; every wave is through with the K image
      if (kt + 1 < nkt) img_dma<DH>(Kb, (int)a.ldk, (kt + 1) * KT, Tk, smem, wu, isrc);
    }
    // O^T += V^T P^T
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int ss = 0; ss < NSS; ++ss)
        o[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr<DH>(Vi, ss, ct, lane), pf[ss], o[ct], 0, 0, 0);
    cur ^= 1;
    ATT_PIN(o[NCT - 1][3]);
    ATT_T(5);
  }

  {
    const int qrow = q0 + m;
    if (qrow < Tq) {
      const float inv = li > 0.f ? drop_sc / li : NAN;  // all keys masked -> NaN, as softmax over -inf
      uint16_t* orow = a.out + ((int64_t)sg.q0 + qrow) * a.ldo + h * DH;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) store4(orow + 16 * ct + 4 * g, o[ct], inv);
      if (g == 0) a.lse[(int64_t)z * a.Tq + qrow] = mi * 0.6931471805599453f + __logf(li);  // natural-log units
    }
  }
#ifdef JS2T_ATTN_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_T(7);  // epilogue: output rows + log-sum-exp, stores acknowledged
  if (blockIdx.x == 0 && t == 0)
    for (int i = 0; i < 8; ++i) g_attn_prof[i] = prof_[i];
#endif
}

// ------------------------------------------------------------------------------------------------ dQ
// own rows = queries (like forward); sweeps key tiles; needs K image (row + transposed reads) and V image (row reads)
// dS' of one (own query block, key tile): pv * ((keep ? dp : 0) - delta*(1-p)); the common factor scale/(1-p) is applied to
// dQ once at the end.  lse2 = lse*log2e, scale2 = scale*log2e.  MASKED = some key of the tile is padded / masked out.
// The subtraction rides on the product: the caller starts the dP accumulators at ndl2 = -delta*(1-p), so `dp` arrives as
// dP - delta*(1-p) and a dropped element takes ndl2 itself (one vector instruction per element less).
template <bool MASKED, int NTT, bool REL = false>
__device__ __forceinline__ void dq_elements(f32x4_t (&s)[NTT], const f32x4_t (&dp)[NTT], float scale2, float lse2, float ndl2,
                                            uint32_t bits, uint32_t rowkeym, uint32_t thr, bool drop,
                                            const float* rel_s = nullptr, float* drel_s = nullptr, int key0 = 0, int query = 0, int R = 0,
                                            bool q_live = true, float* edge = nullptr, unsigned long long* drel_i = nullptr) {
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu;
    if (drop) {  // == dropout_keep4_key(dkey, z*Tq + q, col4 = c4base + 4 tt): rowkeym = (row hash + 2 c4base) * M1, see flash_fwd_kernel
      const uint32_t xm = rowkeym + (uint32_t)tt * (8u * HASH32W_M1);
      h0 = hash32w_pre(xm), h1 = hash32w_pre(xm + HASH32W_M1);
    }
    const uint32_t hv[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ri = REL ? rel_index(key0 + 16 * tt + r, query, R) : 0;
      float pv = __builtin_amdgcn_exp2f(fmaf(s[tt][r], scale2, (REL ? rel_s[ri] : 0.f) - lse2));
      if (MASKED) pv = ((bits >> (4 * tt + r)) & 1u) ? pv : 0.f;
      s[tt][r] = pv * (hv[r] >= thr ? dp[tt][r] : ndl2);
      // gradient of the bias = dS (before the 1/(1-p) factor, applied when the block's histogram is flushed)
      // (two thirds of the pairs of a 375-position utterance lie beyond the clipping distance: they all land in the two end
      // bins, which are summed in registers - edge[0] / edge[1] - and posted once per wave; an LDS atomic per pair cost 135 us
      // per layer, twice the rest of the backward pass)
      if (REL && drel_s && q_live) {
        if (ri == 0) edge[0] += s[tt][r];
        else if (ri == 2 * R) edge[1] += s[tt][r];
        else if (s[tt][r] != 0.f) {
          if (drel_i) atomicAdd(&drel_i[ri], js2t_to_fixed(s[tt][r]));  // deterministic mode: integer sums commute
          else atomicAdd(&drel_s[ri], s[tt][r]);
        }
      }
    }
  }
}

// one own-query block of 16 per wave (two need ~350 registers: one wave per SIMD, slower)
// bid / nblk: this block's index among the pass's blocks (the pass is a grid of its own, or a range of the merged grid)
template <int DH, bool REL, bool DROP>
__device__ __forceinline__ void flash_dq_body(const AttnArgs& a, int bid, int nblk) {
  using G = Geo<DH>;
  constexpr int KT = G::KT, NTT = G::NTT, NKS = G::NKS, NCT = G::NCT, NSS = G::NSS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) uint16_t kmask[KMASK_MAX / 16];
  __shared__ float rel_s[REL ? 2 * REL_MAX + 1 : 1], drel_s[REL ? 2 * REL_MAX + 1 : 1];
  __shared__ unsigned long long drel_i[REL ? 2 * REL_MAX + 1 : 1];  // the same histogram in fixed point (deterministic mode)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4, m = lane & 15;
  const int wu = __builtin_amdgcn_readfirstlane(w);  // wave-uniform copy: LDS-DMA destinations stay on the scalar unit
  const ImgSrc<DH> isrc = img_src<DH>(wu, lane);
  const int ntile = (a.Tq + 63) / 64;  // XCD-aware 1-D grid, see flash_fwd_kernel
  const int lid = xcd_remap(bid, nblk);
  const int z = lid / ntile, b = z / a.H, h = z - b * a.H;
  const int q0 = (lid - z * ntile) * 64 + w * 16;
  const Seg sg = seg_of(a, b);
  const int Tq = sg.Tq, Tk = sg.Tk;
  if (a.seg_rows && !a.seg_keys) zero_tail_rows(a, a.dq, a.lddq, a.H * DH, bid, nblk);
  if (q0 - w * 16 >= Tq) return;  // packed rows: a tile behind this utterance's last query (block-uniform, before any barrier)
  const uint16_t* Qb = a.q + (int64_t)sg.q0 * a.ldq + h * DH;
  const uint16_t* Gb = a.d_o + (int64_t)sg.q0 * a.lddo + h * DH;
  const uint16_t* Kb = a.k + (int64_t)sg.k0 * a.ldk + h * DH;
  const uint16_t* Vb = a.v + (int64_t)sg.k0 * a.ldv + h * DH;
  constexpr bool drop = DROP;  // (launched with DROP = dropout_p > 0)
  const float keep_p = 1.f - a.p;
  img_dma<DH>(Kb, (int)a.ldk, 0, Tk, smem, wu, isrc);  // first, as in flash_fwd_kernel
  img_dma<DH>(Vb, (int)a.ldv, 0, Tk, smem + IMG_BYTES, wu, isrc);
  bf16x8_t qf[NKS], gf[NKS];
  float lse2, ndl2;
  {
    own_frags<NKS>(Qb, a.ldq, q0, Tq, lane, qf);
    own_frags<NKS>(Gb, a.lddo, q0, Tq, lane, gf);
    const int qc = min(q0 + m, Tq - 1);
    lse2 = a.lse[(int64_t)z * a.Tq + qc] * 1.4426950408889634f;
    float dsum;
    if (a.dpart) {
      dsum = load_delta<DH>(a, sg.q0, h, z, qc);  // left behind by the product that made dO: nothing to compute, no O to fetch
    } else {
      // delta = rowsum(dO * O) of the own rows: the dO fragments are in registers anyway, O's are fetched once; the four
      // lanes of a row hold a quarter of the head columns each.  Written out for the dK/dV pass, which runs after this one.
      bf16x8_t of[NKS];
      own_frags<NKS>(a.o + (int64_t)sg.q0 * a.ldo + h * DH, a.ldo, q0, Tq, lane, of);
      dsum = 0.f;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)gf[ks][e] * (float)of[ks][e];
      dsum = quad_sum(dsum);
      if (g == 0 && q0 + m < Tq) a.delta[(int64_t)z * a.Tq + q0 + m] = dsum;
    }
    ndl2 = -dsum * keep_p;
  }
  f32x4_t dq[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) dq[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const uint32_t dkey = drop ? dropout_key(a.rng, a.stream) : 0u;
  const float scale2 = a.scale * 1.4426950408889634f;
  const int nkt = (Tk + KT - 1) / KT;
  const bool full_mask = a.mask && a.msq != 0;
  const uint32_t rowkeym0 = (hash32((uint32_t)(z * a.Tq + min(q0 + m, Tq - 1)) ^ dkey) + 2u * (uint32_t)g) * HASH32W_M1;
  const uint32_t thr = (uint32_t)(a.p * 65536.0f);
  stage_kbits<DH>(kmask, a, b, Tk, nkt * KT, t);
  if (REL) {
    stage_rel(rel_s, a, h, t);
    for (int i = t; i < 2 * a.relR + 1; i += 256) drel_s[i] = 0.f, drel_i[i] = 0ull;
  }
  pin_loaded(qf);
  pin_loaded(gf);
  asm volatile("" : "+v"(lse2), "+v"(ndl2));
  int cur = 0;
  float rel_edge[2] = {0.f, 0.f};  // this lane's share of the two end bins of the bias gradient (dq_elements)
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) {
      img_dma<DH>(Kb, (int)a.ldk, (kt + 1) * KT, Tk, smem + (cur ^ 1) * 2 * IMG_BYTES, wu, isrc);
      img_dma<DH>(Vb, (int)a.ldv, (kt + 1) * KT, Tk, smem + (cur ^ 1) * 2 * IMG_BYTES + IMG_BYTES, wu, isrc);
    }
    const unsigned char* Ki = smem + cur * 2 * IMG_BYTES;
    const unsigned char* Vi = Ki + IMG_BYTES;
    const uint32_t kbits = tile_kbits<DH>(kmask, kt, g);
    // with a bias gradient to collect, padded keys beyond Tk must not count: take the masked path for every tile
    const bool tile_clear = !full_mask && !(REL && a.d_rel) && __all(kbits == G::FULL) != 0;  // wave-uniform: every key is live
    bf16x8_t dsf[NSS];
    {
      f32x4_t s[NTT], dp[NTT];
#pragma unroll
      for (int tt = 0; tt < NTT; ++tt) {
        s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dp[tt] = f32x4_t{ndl2, ndl2, ndl2, ndl2};  // dP - delta*(1-p) comes out of the product (dq_elements)
      }
      {  // fragment reads one pair ahead of their products, as in the dK/dV pass
        constexpr int NF = NTT * NKS;
        bf16x8_t fk[2], fv[2];
        fk[0] = img_row<DH>(Ki, 0, 0, lane);
        fv[0] = img_row<DH>(Vi, 0, 0, lane);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          if (i + 1 < NF) {
            fk[(i + 1) & 1] = img_row<DH>(Ki, (i + 1) / NKS, (i + 1) % NKS, lane);
            fv[(i + 1) & 1] = img_row<DH>(Vi, (i + 1) / NKS, (i + 1) % NKS, lane);
          }
          s[i / NKS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[i & 1], qf[i % NKS], s[i / NKS], 0, 0, 0);
          dp[i / NKS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[i & 1], gf[i % NKS], dp[i / NKS], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int i = 0; i < NF - 1; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      const uint32_t rowkeym = rowkeym0 + (uint32_t)kt * ((KT / 2) * HASH32W_M1);  // col4 base (KT/4) kt + g of this tile
      if (tile_clear) {
        dq_elements<false, NTT, REL>(s, dp, scale2, lse2, ndl2, G::FULL, rowkeym, thr, drop, rel_s, nullptr, KT * kt + 4 * g, q0 + m,
                                     a.relR);
      } else {
        const int qc = min(q0 + m, Tq - 1);
        uint32_t bits = kbits;
        if (full_mask) bits = row_kbits<DH>(bits, a.mask + (int64_t)b * a.msb + (int64_t)qc * a.msq, kt, g, Tk);
        dq_elements<true, NTT, REL>(s, dp, scale2, lse2, ndl2, bits, rowkeym, thr, drop, rel_s, a.d_rel ? drel_s : nullptr,
                                    KT * kt + 4 * g, q0 + m, a.relR, q0 + m < Tq, rel_edge, a.d_rel_fix ? drel_i : nullptr);
      }
#pragma unroll
      for (int ss = 0; ss < NSS; ++ss) dsf[ss] = pack8(s[2 * ss], s[2 * ss + 1]);
    }
    // dQ^T += K^T dS^T
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int ss = 0; ss < NSS; ++ss)
        dq[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr<DH>(Ki, ss, ct, lane), dsf[ss], dq[ct], 0, 0, 0);
    cur ^= 1;
  }
  const float dq_sc = a.scale * (drop ? 1.f / keep_p : 1.f);
  if (REL && a.d_rel) {  // the block's histogram of dS over relative distances -> the head's gradient row
    const float e0 = wave_sum(rel_edge[0]), e1 = wave_sum(rel_edge[1]);  // (a lane's own sum and the wave's butterfly: fixed orders)
    const int n = 2 * a.relR + 1;
    if (a.d_rel_fix) {  // deterministic mode: the block's histogram and the head's row as integer sums (1 / (1 - p) at the conversion)
      if (lane == 0) {
        if (e0 != 0.f) atomicAdd(&drel_i[0], js2t_to_fixed(e0));
        if (e1 != 0.f) atomicAdd(&drel_i[2 * a.relR], js2t_to_fixed(e1));
      }
      __syncthreads();
      for (int i = t; i < n; i += 256)
        if (drel_i[i] != 0ull) atomicAdd(a.d_rel_fix + (int64_t)h * n + i, drel_i[i]);
    } else {
      if (lane == 0) {
        if (e0 != 0.f) atomicAdd(&drel_s[0], e0);
        if (e1 != 0.f) atomicAdd(&drel_s[2 * a.relR], e1);
      }
      __syncthreads();
      const float sc = drop ? 1.f / keep_p : 1.f;
      for (int i = t; i < n; i += 256)
        if (drel_s[i] != 0.f) atomicAdd(a.d_rel + (int64_t)h * n + i, drel_s[i] * sc);
    }
  }
  {
    const int qrow = q0 + m;
    if (qrow < Tq) {
      uint16_t* drow = a.dq + ((int64_t)sg.q0 + qrow) * a.lddq + h * DH;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) store4(drow + 16 * ct + 4 * g, dq[ct], dq_sc);
    }
  }
}

template <int DH, bool REL, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_dq_kernel(AttnArgs a) {
  flash_dq_body<DH, REL, DROP>(a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------ dK / dV
// own rows = keys; sweeps query tiles; Q image (row + transposed reads) and dO image (row + transposed reads)
// One (query tile, own-key block) of the dK/dV pass: probabilities and dS from the S / dP accumulators.
//   pv  = exp2(s * scale*log2e - lse*log2e)        (nlse2 holds -lse*log2e, -inf for rows past Tq -> pv = 0)
//   pd  = keep ? pv : 0                             (dV gets the 1/(1-p) factor once, at the end)
//   ds' = pv * ((keep ? dp : 0) - delta*(1-p))      (dK gets scale/(1-p) once, at the end)
// Both per-query scalars arrive NEGATED (ndl2 = -delta*(1-p)): the caller starts the dP accumulators at ndl2, so `dp` is already
// dP - delta*(1-p) and a dropped element takes ndl2 itself; -lse goes into the fma as it is (two vector instructions per element
// less than negating / subtracting here).
// Dropout: the decision of (query row, key) is the half (key & 1) of hash32w(rowhash + (key >> 1)).  A lane's key is fixed, so
// (i) (key >> 1) * M1 is a per-lane constant and the row hashes arrive pre-multiplied (hash32w_pre: one multiply per word),
// (ii) which half it wants is a per-lane constant, folded into the hash's last multiply: mulsel = M2 for the high half,
// M2 << 16 for the low one (x * (M2 << 16) = (the low half of x * M2) << 16), and the decision is one unsigned compare of the
// product with thr << 16 - no extraction per element.  (A first version shared each word between the two lanes of a key pair -
// two hashes + two DPP moves + four selects per four elements: more instructions than the four hashes.)
template <bool KEYCHECK, bool FULLMASK, int NTT, bool REL = false>
__device__ __forceinline__ void dkv_elements(f32x4_t (&s)[NTT], f32x4_t (&dp)[NTT], const float* nlse2, const float* ndl2,
                                             const uint32_t* rkp, float scale2, bool key_ok, int key, uint32_t thr, bool drop, int g,
                                             int par, const uint8_t* mcol, int64_t msq, int qbase, int Tq, const float* rel_s = nullptr,
                                             int R = 0) {
#pragma unroll
  for (int tt = 0; tt < NTT; ++tt) {
    const f32x4_t l4 = *(const f32x4_t*)(nlse2 + 16 * tt + 4 * g);
    const f32x4_t d4 = *(const f32x4_t*)(ndl2 + 16 * tt + 4 * g);
    uint32_t h[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (drop) {
      const uint4 rk4 = *(const uint4*)(rkp + 16 * tt + 4 * g);  // (row hash) * M1 of this lane's four rows
      const uint32_t c2m = (uint32_t)(key >> 1) * HASH32W_M1;     // per-lane constant: hoisted out of the sweep
      const uint32_t mulsel = par ? HASH32W_M2 : HASH32W_M2 << 16;  // (par = key & 1)
      h[0] = hash32w_pre(rk4.x + c2m, mulsel), h[1] = hash32w_pre(rk4.y + c2m, mulsel);
      h[2] = hash32w_pre(rk4.z + c2m, mulsel), h[3] = hash32w_pre(rk4.w + c2m, mulsel);
    }
    const uint32_t thr_hi = thr << 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float pv = __builtin_amdgcn_exp2f(
          REL ? fmaf(s[tt][r], scale2, rel_s[rel_index(key, qbase + 16 * tt + 4 * g + r, R)] + l4[r]) : fmaf(s[tt][r], scale2, l4[r]));
      if (KEYCHECK) pv = key_ok ? pv : 0.f;
      if (FULLMASK) {
        const int qrow = min(qbase + 16 * tt + 4 * g + r, Tq - 1);
        pv = mcol[(int64_t)qrow * msq] ? pv : 0.f;
      }
      const bool keep = h[r] >= thr_hi;  // the wanted half sits in the upper 16 bits (thr <= 0xffff; no dropout: h = all ones, thr = 0)
      s[tt][r] = keep ? pv : 0.f;
      dp[tt][r] = pv * (keep ? dp[tt][r] : d4[r]);
    }
  }
}

// one own-key block of 16 per wave: with two (32 keys per wave) the dK / dV accumulators, the own K / V fragments and the
// S / dP tiles need > 400 registers - one wave per SIMD, nothing to overlap the MFMA, exp / dropout VALU work and LDS
// latency with.  One block per wave fits 256 registers, i.e. two waves per SIMD.
template <int DH, bool REL, bool DROP>
__device__ __forceinline__ void flash_dkv_body(const AttnArgs& a, int bid, int nblk) {
  using G = Geo<DH>;
  constexpr int KT = G::KT, NTT = G::NTT, NKS = G::NKS, NCT = G::NCT, NSS = G::NSS;  // KT = queries per swept tile here
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) float lse_s[2][KT], dl_s[2][KT];
  __shared__ __attribute__((aligned(16))) uint32_t rk_s[2][KT];
  __shared__ float rel_s[REL ? 2 * REL_MAX + 1 : 1];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4, m = lane & 15;
  const int wu = __builtin_amdgcn_readfirstlane(w);  // wave-uniform copy: LDS-DMA destinations stay on the scalar unit
  const ImgSrc<DH> isrc = img_src<DH>(wu, lane);
  const int ntile = (a.Tk + 63) / 64;  // XCD-aware 1-D grid: the key tiles of a head share its Q / dO
  const int lid = xcd_remap(bid, nblk);
  const int z = lid / ntile, b = z / a.H, h = z - b * a.H;
  const int k0 = (lid - z * ntile) * 64 + w * 16;
  const Seg sg = seg_of(a, b);
  const int Tq = sg.Tq, Tk = sg.Tk;
  if (a.seg_rows) {
    zero_tail_rows(a, a.dk, a.lddk, a.H * DH, bid, nblk);
    zero_tail_rows(a, a.dv, a.lddv, a.H * DH, bid, nblk);
  }
  if (k0 - w * 16 >= Tk) return;  // packed rows: a tile behind this utterance's last key (block-uniform, before any barrier)
  const uint16_t* Qb = a.q + (int64_t)sg.q0 * a.ldq + h * DH;
  const uint16_t* Gb = a.d_o + (int64_t)sg.q0 * a.lddo + h * DH;
  const uint16_t* Kb = a.k + (int64_t)sg.k0 * a.ldk + h * DH;
  const uint16_t* Vb = a.v + (int64_t)sg.k0 * a.ldv + h * DH;
  img_dma<DH>(Qb, (int)a.ldq, 0, Tq, smem, wu, isrc);  // first, as in flash_fwd_kernel: the key-validity test below waits for its mask byte
  img_dma<DH>(Gb, (int)a.lddo, 0, Tq, smem + IMG_BYTES, wu, isrc);
  bf16x8_t kf[NKS], vf[NKS];
  own_frags<NKS>(Kb, a.ldk, k0, Tk, lane, kf);
  own_frags<NKS>(Vb, a.ldv, k0, Tk, lane, vf);
  f32x4_t dk[NCT], dv[NCT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    dk[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    dv[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  constexpr bool drop = DROP;  // (launched with DROP = dropout_p > 0)
  const uint32_t dkey = drop ? dropout_key(a.rng, a.stream) : 0u;
  const float keep_p = 1.f - a.p;  // delta is pre-multiplied by it so that dS carries a common 1/(1-p)
  const uint32_t thr = (uint32_t)(a.p * 65536.0f);
  const float scale2 = a.scale * 1.4426950408889634f;
  const bool full_mask = a.mask && a.msq != 0;
  const int key = k0 + m;
  // key-padding masks depend on the own key only: resolved once, outside the query sweep
  const bool key_valid = key < Tk && (!a.mask || a.msq != 0 || a.mask[(int64_t)b * a.msb + key] != 0);
  const bool all_keys = __all(key_valid) != 0;  // wave-uniform: no per-element key test needed
  const int nqt = (Tq + KT - 1) / KT;
  if (REL) stage_rel(rel_s, a, h, t);  // visible after the first barrier of the sweep
  // per-query scalars of the NEXT tile (log-sum-exp, delta) travel one iteration ahead in registers: fetching
  // them at the top of the iteration they are used in put a global-load round trip in front of every tile
  float lse_r = 0.f, dl_r = 0.f;
  if (t < KT) {
    const int qc = min(t, Tq - 1);
    lse_r = a.lse[(int64_t)z * a.Tq + qc];
    dl_r = load_delta<DH>(a, sg.q0, h, z, qc);
  }
  pin_loaded(kf);
  pin_loaded(vf);
  int cur = 0;
#ifdef JS2T_ATTN_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  for (int qt = 0; qt < nqt; ++qt) {
    if (t < KT) {
      const bool live = qt * KT + t < Tq;
      lse_s[cur][t] = live ? -lse_r * 1.4426950408889634f : -INFINITY;  // negated (dkv_elements); rows past Tq: exp2(-inf) = 0
      dl_s[cur][t] = -dl_r * keep_p;
      rk_s[cur][t] = hash32((uint32_t)(z * a.Tq + min(qt * KT + t, Tq - 1)) ^ dkey) * HASH32W_M1;  // pre-multiplied: hash32w_pre
    }
    ATT_T(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATT_T(1);
    __syncthreads();
    ATT_T(2);
    // the next tile's images are requested in pairs under this tile's products; behind the last tile the requests go on (tile 0
    // again, into the idle buffer: nobody reads it) - a branch around them would cut the product loops into blocks
    const bool more = qt + 1 < nqt;
    const int nrow0 = more ? (qt + 1) * KT : 0;
    unsigned char* nQ = smem + (cur ^ 1) * 2 * IMG_BYTES;
    auto request = [&](const uint16_t* base, int ld, unsigned char* img, int q) {
      img_dma_piece<DH>(base, ld, nrow0, Tq, img, wu, isrc, q);
      img_dma_piece<DH>(base, ld, nrow0, Tq, img, wu, isrc, q + 1);
    };
    if (more && t < KT) {
      const int qc = min((qt + 1) * KT + t, Tq - 1);
      lse_r = a.lse[(int64_t)z * a.Tq + qc];
      dl_r = load_delta<DH>(a, sg.q0, h, z, qc);
    }
    ATT_T(3);
    const unsigned char* Qi = smem + cur * 2 * IMG_BYTES;
    const unsigned char* Gi = Qi + IMG_BYTES;
    // S = Q K^T and dP = dO V^T with D rows = tile queries, D cols = own keys:
    // s[tt][r] <-> (query KT*qt + 16tt + 4g + r, key k0 + m)
    bf16x8_t pf[NSS], dsf[NSS];
#pragma unroll
    for (int hf = 0; hf < NTT / 4; ++hf) {  // 64 queries at a time (register budget: S and dP of 128 queries spill)
      f32x4_t s[4], dp[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        s[tt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dp[tt] = *(const f32x4_t*)(dl_s[cur] + 64 * hf + 16 * tt + 4 * g);  // -delta*(1-p) of the element's query: the product adds dP
      }
      // every product takes a fresh 1 KB fragment from LDS (a wave owns 16 keys: nothing is reused): the reads run ONE PAIR AHEAD
      // of the products, pinned by scheduling groups - left alone the compiler (230 registers) issued read, wait, product, 32 times
      bf16x8_t fq[2], fg[2];
      fq[0] = img_row<DH>(Qi, 4 * hf, 0, lane);
      fg[0] = img_row<DH>(Gi, 4 * hf, 0, lane);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const int i = tt * NKS + ks, nt = (i + 1) / NKS, nks = (i + 1) % NKS;
          if (i + 1 < 4 * NKS) {
            fq[(i + 1) & 1] = img_row<DH>(Qi, 4 * hf + nt, nks, lane);
            fg[(i + 1) & 1] = img_row<DH>(Gi, 4 * hf + nt, nks, lane);
          }
          s[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[i & 1], kf[ks], s[tt], 0, 0, 0);
          dp[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fg[i & 1], vf[ks], dp[tt], 0, 0, 0);
        }
        if (hf == 0 && (tt & 1) == 0) request(Qb, (int)a.ldq, nQ, tt);  // pieces 0-1 behind tt = 0, 2-3 behind tt = 2
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int i = 0; i < 4 * NKS - 1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the next pair of fragments
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // this pair's products
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      const uint8_t* mcol = full_mask ? a.mask + (int64_t)b * a.msb + min(key, Tk - 1) : nullptr;
      const float* lp = lse_s[cur] + 64 * hf;
      const float* dlp = dl_s[cur] + 64 * hf;
      const uint32_t* rp = rk_s[cur] + 64 * hf;
      const int qbase = qt * KT + 64 * hf;
      if (full_mask)
        dkv_elements<true, true, 4, REL>(s, dp, lp, dlp, rp, scale2, key_valid, key, thr, drop, g, m & 1, mcol, a.msq, qbase, Tq, rel_s,
                                         a.relR);
      else if (!all_keys)
        dkv_elements<true, false, 4, REL>(s, dp, lp, dlp, rp, scale2, key_valid, key, thr, drop, g, m & 1, nullptr, 0, qbase, Tq, rel_s,
                                          a.relR);
      else
        dkv_elements<false, false, 4, REL>(s, dp, lp, dlp, rp, scale2, true, key, thr, drop, g, m & 1, nullptr, 0, qbase, Tq, rel_s,
                                           a.relR);
      pf[2 * hf] = pack8(s[0], s[1]);
      pf[2 * hf + 1] = pack8(s[2], s[3]);
      dsf[2 * hf] = pack8(dp[0], dp[1]);
      dsf[2 * hf + 1] = pack8(dp[2], dp[3]);
    }
    ATT_PIN(dsf[NSS - 1]);
    ATT_T(4);
    // dV^T += dO^T Pd ; dK^T += Q^T dS   (contraction over the tile's queries)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
      for (int ss = 0; ss < NSS; ++ss) {
        dv[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr<DH>(Gi, ss, ct, lane), pf[ss], dv[ct], 0, 0, 0);
        dk[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr<DH>(Qi, ss, ct, lane), dsf[ss], dk[ct], 0, 0, 0);
      }
      if (ct == 0 || ct == NCT / 2) request(Gb, (int)a.lddo, nQ + IMG_BYTES, ct == 0 ? 0 : 2);
    }
    cur ^= 1;
    ATT_PIN(dk[NCT - 1][3]);
    ATT_T(5);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests behind the last tile
#ifdef JS2T_ATTN_PROF
  if (bid == 0 && t == 0)
    for (int i = 0; i < 8; ++i) g_attn_prof[i] = prof_[i];
#endif
  const float dv_sc = drop ? 1.f / keep_p : 1.f, dk_sc = a.scale * dv_sc;
  if (key < Tk) {
    uint16_t* krow = a.dk + ((int64_t)sg.k0 + key) * a.lddk + h * DH;
    uint16_t* vrow = a.dv + ((int64_t)sg.k0 + key) * a.lddv + h * DH;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      store4(krow + 16 * ct + 4 * g, dk[ct], dk_sc);
      store4(vrow + 16 * ct + 4 * g, dv[ct], dv_sc);
    }
  }
}

template <int DH, bool REL, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_dkv_kernel(AttnArgs a) {
  flash_dkv_body<DH, REL, DROP>(a, blockIdx.x, gridDim.x);
}

// Both passes as ONE grid: two grids of 768 blocks (an encoder layer) on 512 resident slots are 2 x 1.5 rounds, one of 1536 is
// three; on the decoder's small grids the passes run side by side.  Needs delta before either pass starts, i.e. the partial sums
// of js2t_attn_desc.delta_partial (handing delta from the dQ blocks to the dK/dV blocks inside one grid costs more than the
// merged grid saves: profiles/README.md, round 2).  The dK/dV blocks (the longer ones) go first; their range is padded to a
// multiple of 8 blocks, which keeps the XCD-aware order of both ranges.
template <int DH, bool REL, bool DROP>
__global__ __launch_bounds__(256, 2) void flash_bwd_kernel(AttnArgs a, int n_dkv, int n_dkv_pad) {
  // n_dkv_pad = n_dkv rounded up to a multiple of 8: up to seven idle blocks between the ranges keep the dQ range's block ids
  // congruent to their XCDs (a batch of 33 utterances x 4 heads x 7 tiles used to fall back to two launches)
  if ((int)blockIdx.x < n_dkv) flash_dkv_body<DH, REL, DROP>(a, blockIdx.x, n_dkv);
  else if ((int)blockIdx.x >= n_dkv_pad) flash_dq_body<DH, REL, DROP>(a, (int)blockIdx.x - n_dkv_pad, (int)gridDim.x - n_dkv_pad);
}

template <typename K>
int set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    js2t_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
    return JS2T_ERR_LAUNCH;
  }
  return JS2T_OK;
}

int check_common(const js2t_attn_desc* d) {
  JS2T_CHECK(d != nullptr, "flash_attn: null descriptor");
  JS2T_CHECK(d->head_dim == 128 || d->head_dim == 64, "flash_attn: head size %d not supported (128 or 64; use the unfused path)",
             d->head_dim);
  JS2T_CHECK(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "flash_attn: bad sizes");
  JS2T_CHECK(d->Tk <= KMASK_MAX - 128, "flash_attn: at most %d keys", KMASK_MAX - 128);
  JS2T_CHECK((int64_t)d->B * d->H <= 65535, "flash_attn: B*H too large");
  JS2T_CHECK(d->q && d->k && d->v && d->lse, "flash_attn: null pointer");
  JS2T_CHECK((d->ldq % 8) == 0 && (d->ldk % 8) == 0 && (d->ldv % 8) == 0, "flash_attn: leading dims must be multiples of 8");
  JS2T_CHECK(((((uintptr_t)d->q) | ((uintptr_t)d->k) | ((uintptr_t)d->v)) & 15) == 0, "flash_attn: q/k/v must be 16-byte aligned");
  JS2T_CHECK(d->ldq < (1 << 17) && d->ldk < (1 << 17) && d->ldv < (1 << 17) && d->ld_do < (1 << 17),
             "flash_attn: leading dims must be below 131072 (32-bit tile offsets)");
  JS2T_CHECK(d->dropout_p >= 0.f && d->dropout_p < 1.f && (d->dropout_p == 0.f || d->rng_state), "flash_attn: bad dropout args");
  JS2T_CHECK(!d->rel_bias || (d->rel_R >= 1 && d->rel_R <= REL_MAX), "flash_attn: rel_R must be 1..%d", REL_MAX);
  JS2T_CHECK(!d->d_rel_bias || d->rel_bias, "flash_attn: d_rel_bias without rel_bias");
  JS2T_CHECK(!d->seg || d->seg_keys || d->Tq == d->Tk, "flash_attn: packed rows (seg) are for self-attention, Tq == Tk = the longest entry");
  JS2T_CHECK(!d->seg_keys || d->seg, "flash_attn: seg_keys without seg");
  JS2T_CHECK(!d->seg || (d->seg_rows >= 0 && d->seg_rows < (1ll << 31) && ((d->H * d->head_dim) & 3) == 0), "flash_attn: bad seg_rows");
  return JS2T_OK;
}

AttnArgs to_args(const js2t_attn_desc* d) {
  AttnArgs a;
  a.q = (const uint16_t*)d->q; a.k = (const uint16_t*)d->k; a.v = (const uint16_t*)d->v;
  a.o = (const uint16_t*)d->o; a.d_o = (const uint16_t*)d->d_o; a.out = (uint16_t*)d->o;
  a.dq = (uint16_t*)d->dq; a.dk = (uint16_t*)d->dk; a.dv = (uint16_t*)d->dv;
  a.lse = d->lse; a.delta = d->delta; a.mask = d->mask;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv; a.ldo = d->ldo; a.lddo = d->ld_do; a.lddq = d->ld_dq; a.lddk = d->ld_dk;
  a.lddv = d->ld_dv; a.msb = d->mask_sb; a.msq = d->mask_sq;
  a.B = d->B; a.H = d->H; a.Tq = d->Tq; a.Tk = d->Tk;
  a.scale = d->scale; a.p = d->dropout_p; a.rng = d->rng_state; a.stream = d->rng_stream;
  a.rel = d->rel_bias; a.d_rel = d->d_rel_bias; a.relR = d->rel_bias ? d->rel_R : 0;
  a.dpart = d->delta_partial; a.dgroups = d->delta_groups;
  a.seg = d->seg;
  a.seg_rows = d->seg ? (int)d->seg_rows : 0;
  a.seg_keys = d->seg ? d->seg_keys : 0;
  a.d_rel_fix = nullptr;
  return a;
}

int g_attn_fwd_sb = -1;  // -1: by shape, 0 / 1: forced (js2t_debug_attn_fwd_sb; measurements)
bool g_attn_bwd_merge = true;  // js2t_debug_attn_bwd_merge(0): the two passes as two launches also with delta_partial (A/B)
template <int DH, bool REL, bool DROP>
int launch_fwd(const js2t_attn_desc* d, hipStream_t s) {
  static bool once = false;
  if (!once) {
    int rc = set_lds(flash_fwd_kernel<DH, REL, false, DROP>, 4 * IMG_BYTES);
    if (rc) return rc;
    rc = set_lds(flash_fwd_kernel<DH, REL, true, DROP>, 2 * IMG_BYTES + KMASK_MAX / 8);
    if (rc) return rc;
    once = true;
  }
  AttnArgs a = to_args(d);
  const int nblk = cdiv(d->Tq, 64) * d->B * d->H;
  constexpr int KT = Geo<DH>::KT;
  const int kmask_bytes = (cdiv(d->Tk, KT) * KT / 8 + 15) & ~15;  // one 16-bit word per 16 keys
  // three single-buffered blocks per CU when the grid does not fit two per CU (and the key mask leaves room for three)
  const bool sb = g_attn_fwd_sb >= 0 ? g_attn_fwd_sb != 0 : nblk > 512;
  if (sb) hipLaunchKernelGGL((flash_fwd_kernel<DH, REL, true, DROP>), dim3(nblk), dim3(256), 2 * IMG_BYTES + kmask_bytes, s, a);
  else hipLaunchKernelGGL((flash_fwd_kernel<DH, REL, false, DROP>), dim3(nblk), dim3(256), 4 * IMG_BYTES, s, a);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

template <int DH, bool REL, bool DROP>
int launch_bwd(const js2t_attn_desc* d, hipStream_t s) {
  static bool once = false;
  if (!once) {
    int rc = set_lds(flash_dq_kernel<DH, REL, DROP>, 4 * IMG_BYTES);
    if (rc) return rc;
    rc = set_lds(flash_dkv_kernel<DH, REL, DROP>, 4 * IMG_BYTES);
    if (rc) return rc;
    rc = set_lds(flash_bwd_kernel<DH, REL, DROP>, 4 * IMG_BYTES);
    if (rc) return rc;
    once = true;
  }
  AttnArgs a = to_args(d);
  const int n_dq = cdiv(d->Tq, 64) * d->B * d->H, n_dkv = cdiv(d->Tk, 64) * d->B * d->H;
  // deterministic mode (js2t_set_deterministic): the bias gradient's histogram as integer sums in a library scratch, converted
  // (x 1 / (1 - p)) and added into d_rel_bias behind the kernels
  const int64_t n_fix = (REL && g_js2t_deterministic && d->d_rel_bias) ? (int64_t)d->H * (2 * d->rel_R + 1) : 0;
  if (n_fix) {
    long long* fix = js2t_fixed_scratch((size_t)n_fix, s);
    if (!fix) return JS2T_ERR_INVALID;  // (the error string says why)
    JS2T_CHECK(hipMemsetAsync(fix, 0, (size_t)n_fix * sizeof(long long), s) == hipSuccess, "flash_attn_bwd: memset failed");
    a.d_rel_fix = (unsigned long long*)fix;
  }
  if (d->delta_partial && g_attn_bwd_merge) {
    const int n_dkv_pad = (n_dkv + 7) & ~7;
    hipLaunchKernelGGL((flash_bwd_kernel<DH, REL, DROP>), dim3(n_dkv_pad + n_dq), dim3(256), 4 * IMG_BYTES, s, a, n_dkv, n_dkv_pad);
    JS2T_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL((flash_dq_kernel<DH, REL, DROP>), dim3(n_dq), dim3(256), 4 * IMG_BYTES, s, a);  // also writes delta
    JS2T_LAUNCH_CHECK();
    hipLaunchKernelGGL((flash_dkv_kernel<DH, REL, DROP>), dim3(cdiv(d->Tk, 64) * d->B * d->H), dim3(256), 4 * IMG_BYTES, s, a);
    JS2T_LAUNCH_CHECK();
  }
  if (n_fix) return js2t_fixed_to_float_add((const long long*)a.d_rel_fix, d->d_rel_bias, n_fix, d->dropout_p > 0.f ? 1.f / (1.f - d->dropout_p) : 1.f, s);
  return JS2T_OK;
}

}  // namespace

#ifdef JS2T_ATTN_PROF
extern "C" int js2t_debug_attn_prof(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_attn_prof), 64); }
#endif
extern "C" void js2t_debug_attn_fwd_sb(int mode) { g_attn_fwd_sb = mode; }
extern "C" void js2t_debug_attn_bwd_merge(int on) { g_attn_bwd_merge = on != 0; }
extern "C" int js2t_flash_attn_fwd(const js2t_attn_desc* d, js2t_stream stream) {
  int rc = check_common(d);
  if (rc) return rc;
  JS2T_CHECK(d->o && (d->ldo % 4) == 0 && ((((uintptr_t)d->o)) & 7) == 0, "flash_attn_fwd: bad output");
  hipStream_t s = (hipStream_t)stream;
  if (d->dropout_p > 0.f) {
    if (d->rel_bias) return d->head_dim == 128 ? launch_fwd<128, true, true>(d, s) : launch_fwd<64, true, true>(d, s);
    return d->head_dim == 128 ? launch_fwd<128, false, true>(d, s) : launch_fwd<64, false, true>(d, s);
  }
  if (d->rel_bias) return d->head_dim == 128 ? launch_fwd<128, true, false>(d, s) : launch_fwd<64, true, false>(d, s);
  return d->head_dim == 128 ? launch_fwd<128, false, false>(d, s) : launch_fwd<64, false, false>(d, s);
}

extern "C" int js2t_flash_attn_bwd(const js2t_attn_desc* d, js2t_stream stream) {
  int rc = check_common(d);
  if (rc) return rc;
  JS2T_CHECK(d->d_o && d->dq && d->dk && d->dv && (d->delta_partial || (d->o && d->delta)), "flash_attn_bwd: null pointer");
  JS2T_CHECK(!d->delta_partial || d->delta_groups == d->H * d->head_dim / 64,
             "flash_attn_bwd: delta_partial holds H * head_dim / 64 groups per row");
  JS2T_CHECK((d->ld_do % 8) == 0 && (d->ld_dq % 4) == 0 && (d->ld_dk % 4) == 0 && (d->ld_dv % 4) == 0 && (d->ldo % 8) == 0,
             "flash_attn_bwd: bad leading dims");
  JS2T_CHECK(((((uintptr_t)d->d_o) | ((uintptr_t)d->o)) & 15) == 0 &&
                 ((((uintptr_t)d->dq) | ((uintptr_t)d->dk) | ((uintptr_t)d->dv)) & 7) == 0,
             "flash_attn_bwd: misaligned gradient buffers");
  hipStream_t s = (hipStream_t)stream;
  if (d->dropout_p > 0.f) {
    if (d->rel_bias) return d->head_dim == 128 ? launch_bwd<128, true, true>(d, s) : launch_bwd<64, true, true>(d, s);
    return d->head_dim == 128 ? launch_bwd<128, false, true>(d, s) : launch_bwd<64, false, true>(d, s);
  }
  if (d->rel_bias) return d->head_dim == 128 ? launch_bwd<128, true, false>(d, s) : launch_bwd<64, true, false>(d, s);
  return d->head_dim == 128 ? launch_bwd<128, false, false>(d, s) : launch_bwd<64, false, false>(d, s);
}
