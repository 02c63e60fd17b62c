/*
 * joeys2t_hip.h — C ABI of libjoeys2t_hip.so, the MI355X (gfx950 / CDNA4) kernel library behind the
 * JoeyS2T speech-to-text hot path.
 *
 * The reference (may-/joeys2t) is pure Python: its "native layer" is the set of ATen / cuDNN / torchaudio
 * operators its modules call.  Each entry point below replaces one of those operator call sites; the
 * reference file:line it stands in for is cited on every declaration (paths relative to the reference
 * repo root).  The Python host side (package joeys2t_amd) binds these with ctypes — see INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless a parameter says "host";
 *  - the caller allocates every output and workspace; the library never allocates, frees or keeps a
 *    pointer after the call returns;
 *  - work is enqueued on `stream` and never synchronised;
 *  - return value 0 = enqueued, negative = rejected (nothing enqueued); js2t_last_error() gives the
 *    thread-local message;
 *  - tensors are dense row-major unless strides are passed; "dt" arguments are JS2T_F32 / JS2T_BF16
 *    (bf16 carried as raw 16-bit words); all arithmetic accumulates in f32.
 */
#ifndef JOEYS2T_HIP_H
#define JOEYS2T_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* js2t_stream; /* hipStream_t */

enum { JS2T_F32 = 0, JS2T_BF16 = 1, JS2T_FP8_E4M3 = 2 /* OCP e4m3fn bytes: js2t_gemm operands, js2t_quantize_fp8 output */ };
enum { JS2T_ACT_NONE = 0, JS2T_ACT_RELU = 1, JS2T_ACT_GELU = 2, JS2T_ACT_SWISH = 3, JS2T_ACT_TANH = 4, JS2T_ACT_HARDSWISH = 5 };
enum { JS2T_OK = 0, JS2T_ERR_INVALID = -1, JS2T_ERR_LAUNCH = -2, JS2T_ERR_UNSUPPORTED = -3 };

/* Library identity / error channel. */
int js2t_abi_version(void);
const char* js2t_last_error(void);

/* --------------------------------------------------------------------------------------------------
 * GEMM with fused epilogue (MFMA).  Replaces every nn.Linear / torch.matmul / nn.Conv1d call on the
 * path: transformer_layers.py:75-77,89,102,107 (q/k/v/out projections, QK^T, PV),
 * transformer_layers.py:147-153 (position-wise feed-forward), decoders.py:620-623 (vocabulary and CTC
 * projections), encoders.py:338-346,364-366 (Conv1d k=5,s=2 as implicit GEMM) and their autograd
 * backward passes.
 *
 *   acc(m,n)  = sum_k A(m,k) * B(n,k)                      (f32 accumulate)
 *   v         = alpha * [*alpha_dev] * acc + bias[n]
 *   [preact(m,n) = v]           v = act(v)
 *   v         = dropout(v)       (Philox keep-mask over (m,n), scaled 1/(1-p))
 *   v        += res_scale * residual(m,n)
 *   v         = gate(m,n) > 0 ? v * gate_scale : 0         (ReLU/dropout backward gate)
 *   C(m,n)    = v + beta * C(m,n)
 *
 * Operand addressing:  A(m,k) = A[m*lda + k] (trans_a=0) or A[k*lda + m] (trans_a=1);
 *                      B(n,k) = B[n*ldb + k] (trans_b=0, i.e. an nn.Linear weight) or B[k*ldb + n].
 * Batched: problem z in [0,batch) splits as zo = z / batch_inner, zi = z % batch_inner and every
 * operand pointer advances by zo*stride_o + zi*stride_i (elements) — this is how [B,T,H,dh] heads are
 * addressed without transposes.
 * conv != 0 turns A into an implicit im2col view of x[Bc, conv_tin, conv_c]: the A index that runs
 * along lda ("row" for trans_a=0, "k" for trans_a=1) is (b*conv_tout + t), the contiguous index is
 * kw*conv_c + c, and A = x[b, t*conv_stride - conv_pad + kw, c] (0 outside [0,conv_tin)).
 * dtype_ab = BF16 needs 16-byte aligned operands and leading dimensions that are multiples of 8.
 * split_k > 1 (weight gradients: few output tiles, K = tokens) spreads the reduction over split_k blocks per tile.
 * LayerNorm fold (pre-LN blocks of transformer_layers.py:267-289,348-407; bf16, model width 512): nn.LayerNorm in front of a
 * block's first nn.Linear is not a kernel of its own.
 *   producer, rs_partial != NULL: the epilogue that WRITES a residual-stream tensor also writes, per row and 64-column group,
 *     the sum and the sum of squares of the values it stores (bf16-rounded) - plain stores, no atomics, nothing to zero.
 *   consumer, ln_partial != NULL (the producer's rs_partial): the product runs on the RAW residual stream x with the weight
 *     js2t_fold_ln_weights derives,
 *       B(n,k) = W(n,k) gamma(k) - (1/K) sum_k' W(n,k') gamma(k')   (gamma-scaled, every row centred: sum_k B(n,k) = 0, so
 *     that sum_k x(m,k) B(n,k) = sum_k (x(m,k) - mean(m)) W(n,k) gamma(k) without the mean ever being subtracted).  At the
 *     start of a tile one lane per row adds the row's eight partial pairs (fixed order: reproducible bit for bit), forms
 *     mean and rstd = 1/sqrt(var + ln_eps), and the epilogue computes
 *       v = rstd(m) * acc(m,n) + bias[n]     with bias = b + W beta     (= LN(x) W^T + b), then activation / dropout as usual;
 *     ln_mean / ln_rstd (optional) receive the statistics for the backward pass.
 * Both: plain k-contiguous bf16 products with a bf16 result (the persistent 192x128 kernel or the 64 / 128-row tile kernel),
 * N % 128 == 0, alpha = 1, a bias; epilogues bias [+ ReLU [+ dropout]] (consumer, K = 512) and bias [+ dropout] + residual
 * (producer, N = 512).
 * a_rowsum != NULL: additionally a_rowsum[m] += sum_k op(A)[m,k] (f32 atomics onto whatever is there).  For a weight
 * gradient dW = dY^T X this is the bias gradient (autograd of nn.Linear, transformer_layers.py:75-107), taken from the
 * dY tiles the product already holds in LDS.  bf16 LDS-DMA path with trans_a = trans_b = 1 only (batch == 1, no conv).
 */
typedef struct js2t_gemm_desc {
  int32_t M, N, K;
  int32_t batch, batch_inner;
  int32_t dtype_ab, dtype_c;
  int32_t trans_a, trans_b;
  const void* A; int64_t lda, a_stride_o, a_stride_i;
  const void* B; int64_t ldb, b_stride_o, b_stride_i;
  void* C;       int64_t ldc, c_stride_o, c_stride_i;
  float alpha;
  const float* alpha_dev;   /* optional device scalar folded into alpha */
  const float* bias;        /* optional f32[N] */
  int32_t act;              /* JS2T_ACT_* */
  void* preact;             /* optional, C's dtype/layout: value before the activation */
  float dropout_p;          /* 0 = off */
  const uint64_t* rng_state;/* device {seed, offset}; required when dropout_p > 0 */
  uint32_t rng_stream;      /* call-site id mixed into the counter */
  const void* residual;     /* optional, C's dtype and batch strides */
  int64_t ldr;
  float res_scale;
  const void* gate;         /* optional, C's dtype and batch strides, leading dim ldg */
  int64_t ldg;
  float gate_scale;
  float beta;
  int32_t conv, conv_tin, conv_tout, conv_c, conv_stride, conv_pad;
  int32_t split_k;          /* >1: K is cut into slices reduced with f32 atomics into a zero-filled f32 C */
  float* a_rowsum;          /* optional f32[M]: += row sums of op(A) */
  /* LayerNorm folded into the product (see below); all optional */
  const float* ln_partial;  /* consumer: f32 [M][8][2], per 64-column group {sum, sum of squares} of A's rows (K = 512) */
  float ln_eps;
  float* ln_mean;           /* consumer out (both or neither): f32 [M] row means and 1/sqrt(var + eps) the epilogue used */
  float* ln_rstd;
  float* rs_partial;        /* producer out: f32 [M][8][2], the same sums over the stored (bf16-rounded) C's rows (N = 512) */
  uint8_t* c8;              /* e4m3 products: optional second output, the result as e4m3 bytes [M][ldc8] (C may then be NULL) ... */
  int64_t ldc8;
  float* c8_state;          /* ... written with the delayed scale c8_state[0]; the launch's max |v| is collected in c8_state[1] */
  const float* c8_mul;      /* ... and *c8_scale_out = c8_state[0] * (*c8_mul or 1): the alpha_dev of the product that consumes c8 */
  float* c8_scale_out;
  float* fp8_state;         /* e4m3 products: delayed-scale state of the kernel that quantised A (js2t_layernorm_fwd_fp8): block 0 hands
                             * the collected maximum over, state[0] = state[1] / 448, state[1] *= 15/16 (decayed, not cleared), when the product is done */
  /* producer out (k-contiguous bf16 products with a plain epilogue, N % 64 == 0): dot_partial[m * (N / 64) + n / 64] = the sum over
   * the 64-column group of bf16(C[m, c]) * dot_src[m, c] (dot_src: bf16 [M, ld_dot]).  The attention output projection's input
   * gradient dO = dY Wo is where dO and the saved attention output O meet first: with dot_src = O the partials are
   * delta = rowsum(dO * O) per head and half head - what the attention backward needs before either of its passes
   * (js2t_attn_desc.delta_partial), so that both run as one grid. */
  const void* dot_src;
  int64_t ld_dot;
  float* dot_partial;
  float* sumsq_partial;     /* js2t_gemm_grouped only (f32 C, no split-K, N % 128 == 0): block b of the launch stores the sum of squares
                             * of the C values it wrote to sumsq_partial[b], b < js2t_gemm_grouped_blocks(M, N, count) - the weight
                             * gradients' share of clip_grad_norm_'s norm (builders.py:68-71) without a pass over them */
} js2t_gemm_desc;

int js2t_gemm(const js2t_gemm_desc* d, js2t_stream stream);
/* `count` independent products of ONE shape (d: M, N, K, leading dimensions, alpha, beta, split_k, dtype_c) whose
 * operands live at unrelated addresses A[i], B[i], C[i] (and optional a_rowsum[i]); d->A/B/C/a_rowsum are ignored.
 * bf16 operands with trans_a = trans_b = 1 and a plain epilogue only: this is the weight-gradient product
 * dW_i = dY_i^T X_i of the nn.Linear layers (autograd of transformer_layers.py:75-107,147-153), deferred to the end
 * of the backward pass and run for all layers of one type in one launch - enough output tiles to fill the chip
 * without cutting K into atomically reduced slices.  Host pointer tables; any count (launched in chunks). */
#define JS2T_GEMM_GROUP_MAX 32
int js2t_gemm_grouped(const js2t_gemm_desc* d, int32_t count, const void* const* A, const void* const* B, void* const* C,
                      float* const* a_rowsum, js2t_stream stream);
/* number of blocks (= entries of sumsq_partial written) of an un-split js2t_gemm_grouped launch */
int64_t js2t_gemm_grouped_blocks(int32_t M, int32_t N, int32_t count);
/* Test hook: when on, bf16 GEMMs use the register-staged kernel (the one implicit-conv operands always use)
 * instead of the LDS-DMA kernel, so both can be checked against each other. */
void js2t_gemm_force_regstage(int on);
/* Test hook: when on, every k-contiguous bf16 product with a bf16 result the 256x256 half-tile-ring kernel can run takes it
 * (by default only products of >= 512 such tiles and K >= 1024 do). */
void js2t_gemm_force_w256(int on);
/* Kernel selection for the persistent 192x128 kernel (k-contiguous bf16 operands, bf16 result, N % 8 == 0, N >= 128,
 * K % 8 == 0, K >= 192, bias / ReLU / dropout / residual-or-gate epilogue) and its reduction-major form inside
 * js2t_gemm_grouped (f32 result, N % 128 == 0, M % 8 == 0, no split-K; only in mode 1): 0 = never, 1 = every product
 * that qualifies (test hook), -1 = k-contiguous products that qualify and have >= 200 tiles (default). */
void js2t_gemm_p192_mode(int mode);
/* Kernel selection inside js2t_gemm_grouped for the 256x128 three-slot-ring kernel (reduction-major bf16 operands, f32 result,
 * M % 256 == 0, N % 128 == 0, 16-byte aligned C; rowsum / sumsq_partial / beta / split_k as the 128x128 kernel, products
 * bit-identical to it - with split_k > 2 to the order of the slices' atomic additions): 0 = never, 1 = every launch that
 * qualifies (test hook), -1 = launches of >= 160 blocks (tiles x members x slices) over K >= 1024 per slice (default: the
 * deferred weight gradients of the encoder and decoder layers, training.py:570-588 of the reference - loss.backward() forms
 * them one by one). */
void js2t_gemm_wg256_mode(int mode);
/* Variant of the persistent 192x128 kernel: 3 = one block per CU with two stages in flight, 2 = two blocks per CU
 * with one stage in flight each (80 KB of LDS per block), 4 = one block per CU of eight multiplying and four
 * requesting waves, anything else = chosen per launch (default: 2 when the product has at least 1.5 tiles per CU,
 * else 4).  All variants produce bit-identical results. */
void js2t_gemm_p192_ring(int nst);
/* Panel-resident kernel (k-contiguous bf16 operands and result, K % 128 == 0, K <= 512, N % 8 == 0, epilogue bias [+ ReLU
 * [+ dropout]] [+ folded LayerNorm] or gate): a 96-column panel of B stays in LDS and only A is streamed, each wave through a
 * ring of its own.  -1 = products with at least 4096 (32-row strip, panel) units (default), 0 = never, 1 = every product that
 * qualifies (test hook).  Results are bit-identical to the persistent 192x128 kernels. */
void js2t_gemm_panel_mode(int mode);

/* --------------------------------------------------------------------------------------------------
 * Element-wise / data-movement kernels.
 */

/* out[i] = a*x[i] + b*y[i] (y may be NULL) — residual-branch gradient sums (transformer_layers.py:283,384,397
 * backward) and loss interpolation scaling (loss.py:164). */
int js2t_axpby(const void* x, float a, const void* y, float b, void* out, int64_t n, int dt, js2t_stream stream);

/* dst[i] = (dst_dt) src[i].  Autocast-style parameter/activation casts (training.py:558 autocast). */
int js2t_cast(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, js2t_stream stream);

/* Weights of the LayerNorm fold (js2t_gemm ln_partial), re-derived from the fp32 masters after every optimizer update: for each
 * table row e = {W, gamma, beta, bias, Wf, bias_f, N, K} (device pointers / sizes as int64):
 *   c[n] = sum_k W[n,k] gamma[k];  Wf[n,k] = bf16(W[n,k] gamma[k] - c[n] / K);  bias_f[n] = bias[n] + sum_k W[n,k] beta[k]
 * (W f32 [N,K], gamma / beta f32 [K], bias f32 [N] or 0, K % 4 == 0).  One launch for all entries. */
int js2t_fold_ln_weights(const int64_t* table, int32_t n_entries, int32_t max_rows, js2t_stream stream);

/* fp8 forward mode (BASELINE.json configs[4]: "fp8 MFMA"; the reference has no fp8 path - joeynmt/config.py:223-225 knows
 * fp16 only - so this is an extension without a parity target).  Per-tensor dynamic scaling, no host sync:
 *   js2t_absmax:        *out = max |x|                                  (out: device float, reset inside the call)
 *   js2t_quantize_fp8:  y = e4m3(x * 448 / *amax), clamped to +-448;  *scale_out = *amax / 448 * (*mul or 1)
 * js2t_gemm with dtype_ab = JS2T_FP8_E4M3 then multiplies the two byte tensors on v_mfma_f32_16x16x32_fp8_fp8 and
 * applies alpha * *alpha_dev (the product of the two scales), bias, ReLU, dropout, residual like the bf16 kernel. */
int js2t_absmax(const void* x, int dt, int64_t n, float* out, js2t_stream stream);
int js2t_quantize_fp8(const void* x, int dt, void* y, int64_t n, const float* amax, const float* mul, float* scale_out,
                      js2t_stream stream);
/* The same in ONE pass with delayed scaling: state f32[4] (16-byte aligned, device) = {scale in use, running max of this
 * call (bits), arrival ticket, unused}.  y = e4m3(clamp(x / state[0])), *scale_out = state[0] * (*mul or 1); the last
 * block to finish replaces state[0] by this call's max |x| / 448 for the NEXT call and clears the counters - a captured
 * hipGraph keeps adapting the scale by itself.  Initialise state = {max|x| / 448 of a calibration tensor, 0, 0, 0}. */
int js2t_quantize_fp8_delayed(const void* x, int dt, void* y, int64_t n, float* state, const float* mul, float* scale_out,
                              js2t_stream stream);

/* y[r,c] = x[r,c] * sigmoid(x[r,c+C]) for x[rows,2C] — F.glu(dim=1) of encoders.py:366 in [B,T,C] layout. */
int js2t_glu_fwd(const void* x, void* y, int64_t rows, int64_t C, int dt, js2t_stream stream);
/* dx[rows,2C] from dy[rows,C] and the saved x. */
int js2t_glu_bwd(const void* x, const void* dy, void* dx, int64_t rows, int64_t C, int dt, js2t_stream stream);
/* The same for x[batch * T, 2C] with a device-resident crop length: positions t >= *valid_t of every batch entry come out as 0
 * (forward) / receive no gradient (backward).  A batch padded to a bucket length (hipGraph replay of varying batches) then
 * presents the following convolution with what the reference's cropped tensor + nn.Conv1d zero padding present
 * (encoders.py:356-368).  valid_t == NULL: no crop. */
int js2t_glu_fwd_crop(const void* x, void* y, int64_t rows, int64_t C, int64_t T, const int64_t* valid_t, int dt, js2t_stream stream);
int js2t_glu_bwd_crop(const void* x, const void* dy, void* dx, int64_t rows, int64_t C, int64_t T, const int64_t* valid_t, int dt,
                      js2t_stream stream);

/* Transposed bf16 shadows of the 2-D weights: for every group g (table[g] = {element offset, rows, cols, first 64x64 tile},
 * int64 on the device) the [rows, cols] block at src + offset is written as [cols, rows] at dst + offset.  With W^T at
 * hand the input gradient dX = dY W of every nn.Linear (autograd of transformer_layers.py:75-107,147-153) is a product of
 * two k-contiguous operands, i.e. it takes the faster mainloop and the register-direct epilogue of js2t_gemm. */
int js2t_transpose_groups(const void* src, void* dst, const int64_t* table, int32_t n_groups, int64_t total_tiles, js2t_stream stream);

/* y[b,t,:] = dropout(x[b,t,:] + pe[t,:] (+ extra[b,t,:])) — PositionalEncoding.forward
 * (transformer_layers.py:204-213) + emb_dropout (encoders.py:273-276, decoders.py:599-602).
 * pe is f32[>=T, D] or NULL (plain dropout: ConformerEncoder's emb_dropout after its input Linear, encoders.py:433-435);
 * extra (prompt-mask embedding) may be NULL. */
int js2t_add_pe_dropout(const void* x, const float* pe, const void* extra, void* y, int64_t B, int64_t T,
                        int64_t D, int dt, float p, const uint64_t* rng_state, uint32_t rng_stream,
                        js2t_stream stream);

/* dx = dy * keep(row,col) / (1-p) with the same Philox mask the forward drew for (rng_stream). */
int js2t_dropout_bwd(const void* dy, void* dx, int64_t rows, int64_t cols, int dt, float p,
                     const uint64_t* rng_state, uint32_t rng_stream, js2t_stream stream);

/* dz = scale * dh * act'(z) — backward of the activations of builders.py:24-41 (relu, gelu, swish, tanh).
 * For ReLU, z may be the saved (post-dropout) output: its sign carries both the ReLU and the keep mask. */
int js2t_act_bwd(const void* dh, const void* z, void* dz, int64_t n, int act, int dt, float scale,
                 js2t_stream stream);

/* out[i,:] = table[ids[i],:] * scale — Embeddings.forward (embeddings.py:55-64). */
int js2t_embed_fwd(const int64_t* ids, const void* table, int table_dt, void* out, int out_dt,
                   int64_t n_ids, int64_t D, int64_t vocab, float scale, js2t_stream stream);
/* dtable[ids[i],:] += scale * dout[i,:] (f32 atomics); rows with ids[i]==pad_idx are skipped
 * (nn.Embedding padding_idx, embeddings.py:51). */
int js2t_embed_bwd(const int64_t* ids, const void* dout, int dout_dt, float* dtable, int64_t n_ids,
                   int64_t D, int64_t vocab, float scale, int64_t pad_idx, js2t_stream stream);

/* out[c] = sum_r x[r,c] (f32) — bias gradients of every nn.Linear / nn.Conv1d on the path.
 * partial is a caller-provided f32[colsum_partial_rows(rows) * cols] workspace. */
int64_t js2t_colsum_partial_rows(int64_t rows);
int js2t_colsum(const void* x, int dt, float* out, float* partial, int64_t rows, int64_t cols, int accumulate,
                js2t_stream stream); /* accumulate != 0: out[c] += sum (gradient accumulation in place) */

/* Packed rows of a ragged batch: the encoder stack of a batch whose utterances differ in length runs on the sum(T'_i) live rows
 * (positions behind an utterance's sub-sampled length are dead in the reference: encoders.py:348-373, transformer_layers.py:86-105,
 * loss.py:156-161).  seg: device int32 [B + 1] row offsets (seg[0] = 0).
 *   pack != 0: dst[seg[b] + t, :] = src[b*T + t, :] for t < seg[b+1] - seg[b]; rows seg[B] .. rows_out of dst are zeroed
 *              (the caller's rounding of the packed row count);
 *   pack == 2: only that tail of dst is zeroed (src is not read: pass dst) - buffers the packed kernels leave rows of unwritten;
 *   pack == 0: dst[b*T + t, :] = t < seg[b+1] - seg[b] ? src[seg[b] + t, :] : 0   (rows_out unused).
 * row_bytes % 16 == 0, 16-byte aligned buffers.  Each is the other's adjoint (the backward of pack is unpack and vice versa). */
int js2t_pack_rows(const void* src, void* dst, const int32_t* seg, int32_t B, int32_t T, int64_t rows_out, int64_t row_bytes,
                   int pack, js2t_stream stream);

/* Conv1d weight repack: w[Cout,Cin,K] (torch layout, encoders.py:339-345) <-> wp[Cout, K*Cin] (GEMM layout). */
int js2t_conv_weight_pack(const float* w, void* wp, int wp_dt, int64_t cout, int64_t cin, int64_t k,
                          js2t_stream stream);
int js2t_conv_weight_unpack_grad(const float* dwp_t, float* dw, int64_t cout, int64_t cin, int64_t k,
                                 int accumulate, js2t_stream stream); /* dwp_t is [K*Cin, Cout] (transposed wgrad) */
/* col[(b*tout+t), kw*C+c] = x[b, t*stride-pad+kw, c], zero outside [0, tin): the strided convolution's A operand
 * (nn.Conv1d(k, stride 2, padding k//2) of encoders.py:339-346 as a GEMM) written out once - forward and weight
 * gradient then run on the LDS-DMA kernels, 2.4x faster than gathering the taps inside the GEMM.  16-byte pieces:
 * C % 8 == 0 (bf16) / C % 4 == 0 (f32). */
int js2t_im2col(const void* x, void* col, int64_t B, int64_t tin, int64_t tout, int64_t C, int64_t K,
                int64_t stride, int64_t pad, int dt, js2t_stream stream);
/* dx[b,tau,c] = sum over taps of dcol[(b*tout+t), kw*C+c] with t*stride-pad+kw == tau (conv dgrad gather). */
int js2t_col2im(const void* dcol, void* dx, int64_t B, int64_t tin, int64_t tout, int64_t C, int64_t K,
                int64_t stride, int64_t pad, int dt, js2t_stream stream);

/* out_len = floor((len + 2*(k/2) - (k-1) - 1)/2 + 1) per conv layer — Conv1dSubsampler.get_out_seq_lens_tensor
 * (encoders.py:348-352) followed by lengths_to_padding_mask (helpers.py:459-469): mask[b,t] = t < out_len[b].
 * kernel_sizes is a HOST array. */
int js2t_subsample_lengths_mask(const int64_t* lengths, int64_t* out_lengths, uint8_t* mask, int64_t B,
                                int64_t T_out, const int32_t* kernel_sizes, int32_t n_layers,
                                js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * LayerNorm(eps=1e-6) — transformer_layers.py:146,248,339-340; encoders.py:223-226; decoders.py:549-552.
 * mean/rstd are f32[rows] (saved for backward).  gamma/beta f32[D].
 */
int js2t_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, int64_t rows, int64_t D, float eps, int dt, js2t_stream stream);
/* The same with the result also (y == NULL: only) as e4m3 bytes for the fp8 forward mode (extension, see js2t_quantize_fp8):
 * y8 = e4m3(clamp(LN(x) / S)), S = q_state[0]; the call's max |LN(x)| is collected in q_state[1] (f32[4], 16-byte aligned) and
 * becomes the next call's scale when the consuming product hands it over (js2t_gemm_desc.fp8_state);
 * *q_scale_out = S * (*q_mul or 1).  The quantisation of a LayerNorm-fed nn.Linear input then has no pass of its own. */
int js2t_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y, void* y8, float* q_state,
                           const float* q_mul, float* q_scale_out, float* mean, float* rstd, int64_t rows, int64_t D, float eps,
                           int dt, js2t_stream stream);
/* dx (dt) = LN'(dy) [+ add_scale * add]  (add: optional residual-branch gradient, same shape as dx);
 * dgamma/dbeta f32[D] (may both be NULL); partial = f32[2 * ceil(rows/16) * D] workspace. */
int js2t_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, void* dx, const void* add, float add_scale, float* dgamma,
                       float* dbeta, float* partial, int accumulate, int64_t rows, int64_t D, int dt,
                       js2t_stream stream); /* accumulate != 0: dgamma/dbeta += (in-place gradient accumulation; the vectorised
                                              * kernel adds its block totals with f32 atomics, `partial` is unused) */
/* Same, plus a second output dx_dropped (dt) = js2t_dropout_bwd(dx, drop_p, rng_stream): in a pre-LN layer stack
 * (transformer_layers.py:226-262) the gradient leaving a block's LayerNorm is the gradient entering the previous block,
 * whose backward starts with the dropout mask of its last projection - produced here while dx is still in registers.
 * dx_dropped may be NULL (then identical to js2t_layernorm_bwd); needs D % 8 == 0, D <= 2048, 16-byte aligned rows. */
int js2t_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, const float* mean,
                               const float* rstd, void* dx, const void* add, float add_scale, float* dgamma,
                               float* dbeta, float* partial, int accumulate, int64_t rows, int64_t D, int dt,
                               void* dx_dropped, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                               int32_t acc_copies, int64_t acc_copy_stride, js2t_stream stream);
/* Same, plus a third output n_out (dt) = xhat * gamma + beta - the LayerNorm's forward result, for callers whose forward
 * folded the normalisation into the consuming product (js2t_gemm ln_stats) and never wrote it: the deferred weight gradient
 * dW = dY^T LN(x) reads it.  n_out may be NULL (then identical to js2t_layernorm_bwd_dropout); with n_out, beta is required. */
int js2t_layernorm_bwd_fused(const void* dy, const void* x, const float* gamma, const float* mean,
                             const float* rstd, void* dx, const void* add, float add_scale, float* dgamma,
                             float* dbeta, float* partial, int accumulate, int64_t rows, int64_t D, int dt,
                             void* dx_dropped, float drop_p, const uint64_t* rng_state, uint32_t rng_stream,
                             int32_t acc_copies, int64_t acc_copy_stride, const float* beta, void* n_out, js2t_stream stream);
/* acc_copies > 1 (with accumulate != 0): block b adds its parameter-gradient sums into copy b % acc_copies, i.e. at
 * dgamma / dbeta + (b % acc_copies) * acc_copy_stride floats, instead of every block hammering the same 2 D addresses
 * (same-address atomics serialise: 21.7 -> 12.8 us for 12000 x 512 with 8 copies).  js2t_fold_copies sums the copies of
 * all LayerNorms into the gradients once per step and zeroes them: table = int64[n][3] {workspace offset in floats,
 * destination pointer, count}. */
int js2t_fold_copies(float* ws, const int64_t* table, int32_t n_entries, int32_t copies, int64_t copy_stride,
                     js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * Masked softmax (+ dropout) over attention scores — transformer_layers.py:93-98.
 * S,P,Pd: [Z, Tq, ld] with Z = B*H; mask: uint8, element (b,q,k) at mask[b*mask_sb + q*mask_sq + k]
 * (mask_sq = 0 for a [B,1,Tk] key-padding mask); masked scores act as -inf.  Columns [Tk, ld) of P/Pd
 * are written as zeros so that P can feed the PV GEMM with K = ld.  Pd (dropout applied) may alias P
 * when p == 0.
 */
int js2t_softmax_fwd(const void* S, const uint8_t* mask, void* P, void* Pd, int64_t B, int64_t H,
                     int64_t Tq, int64_t Tk, int64_t ld, int64_t mask_sb, int64_t mask_sq, int dt,
                     float p, const uint64_t* rng_state, uint32_t rng_stream, js2t_stream stream);
/* dS = P * (dP - sum_k dP*P) with dP = dPd * keep/(1-p). */
int js2t_softmax_bwd(const void* P, const void* dPd, void* dS, int64_t Z, int64_t Tq, int64_t Tk,
                     int64_t ld, int dt, float p, const uint64_t* rng_state, uint32_t rng_stream,
                     js2t_stream stream);
/* out[b,q,k] = (1/H) sum_h P[b,h,q,k]  (f32) — head-averaged weights, transformer_layers.py:109-114. */
int js2t_attn_head_mean(const void* P, float* out, int64_t B, int64_t H, int64_t Tq, int64_t Tk,
                        int64_t ld, int dt, js2t_stream stream);

/* Relative-position bias on the materialised attention path (EXTENSION, BASELINE.json configs[4] "rel-pos attn"; the
 * reference's MultiHeadedAttention, transformer_layers.py:86-98, has no such term - this is where it would enter, between the
 * scaled q k^T product and the mask): S[b,h,q,k] += rel_bias[h, clamp(k - q, -R, R) + R], in place on the [B*H, Tq, ld]
 * score buffer; rel_bias f32 [H, 2R+1].  The fused kernels (js2t_flash_attn_*) take the table through js2t_attn_desc. */
int js2t_rel_bias_add(void* S, const float* rel_bias, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t ld, int32_t R,
                      int dt, js2t_stream stream);
/* d_rel_bias[h, r] += sum of dS[b,h,q,k] over all (b, q, k) with clamp(k - q, -R, R) + R == r (dS: gradient of the scaled,
 * biased scores as js2t_softmax_bwd writes it); ADDS into d_rel_bias (f32 [H, 2R+1]). */
int js2t_rel_bias_grad(const void* dS, float* d_rel_bias, int64_t B, int64_t H, int64_t Tq, int64_t Tk, int64_t ld, int32_t R,
                       int dt, js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * Vocabulary-sized rows: log-softmax, label-smoothed cross-entropy, CTC.
 */

/* lse[r] = log sum_v exp(x[r,v]); argmax[r] (optional) = first index of the row maximum.
 * F.log_softmax statistics of model.py:121,126 and search.py:562; argmax of model.py:139-143. */
int js2t_row_lse(const void* x, float* lse, int64_t* argmax, int64_t rows, int64_t V, int dt, js2t_stream stream);

/* CTC best-path decoding of the encoder-side output layer (`return_type="decode_ctc"` hands out ctc_out,
 * model.py:162-166, but nothing in the reference consumes it): best[b,t] = argmax_v logits (js2t_row_lse's argmax);
 * this call collapses repeats and drops blanks inside the first in_len[b] frames: out_ids i64[B,T] (pad-filled),
 * out_len i64[B]. */
int js2t_ctc_collapse(const int64_t* best, const int64_t* in_len, int64_t* out_ids, int64_t* out_len, int64_t B,
                      int64_t T, int64_t blank, int64_t pad, js2t_stream stream);
/* y = x - lse(x) row-wise — F.log_softmax(dim=-1) (model.py:121,126; search.py:258,562). */
int js2t_log_softmax(const void* x, int dt, void* y, int y_dt, int64_t rows, int64_t V, js2t_stream stream);

/* XentLoss on logits (loss.py:16-107 after model.py:121): per row r with gold trg[r]
 *   smoothing > 0: KLDivLoss(sum) against the smoothed target (eps/(V-2) off-gold, 1-eps gold, pad column 0,
 *                  all-zero row when gold == pad) in closed form;  smoothing <= 0: NLLLoss(ignore_index=pad, sum).
 * loss_rows / correct_rows / lse are f32[rows]; correct_rows[r] = 1 if argmax == gold and gold != pad
 * (n_correct of model.py:137-143).  Sum them with js2t_sum_f32. */
int js2t_xent_fwd(const void* logits, int dt, const int64_t* trg, float* loss_rows, float* correct_rows,
                  float* lse, int64_t rows, int64_t V, int64_t pad_idx, float smoothing, js2t_stream stream);
/* dlogits = scale * (*g_dev) * (softmax - target)  (g_dev: optional device scalar = upstream gradient). */
int js2t_xent_bwd(const void* logits, int dt, const int64_t* trg, const float* lse, const float* g_dev,
                  float scale, void* dlogits, int64_t rows, int64_t V, int64_t pad_idx, float smoothing,
                  js2t_stream stream);
/* The same with the gradient stored as out_dt: JS2T_BF16 for f32 logits whose gradient is the bf16 operand of the output layer's
 * backward products next (decoders.py:620 backward) - no f32 gradient [rows, V] to write and cast. */
int js2t_xent_bwd_as(const void* logits, int dt, const int64_t* trg, const float* lse, const float* g_dev, float scale,
                     void* dlogits, int out_dt, int64_t rows, int64_t V, int64_t pad_idx, float smoothing, js2t_stream stream);

/* Running statistics of TrainManager._train_step (training.py:566-586: loss, nll, ctc are normalised by
 * batch.normalize(), batch.py:135-175; n_correct, nseqs, ntokens are counts) in one launch instead of a dozen scalar
 * kernels: stats6 (f64) += [total*inv_norm, nll*inv_norm, ctc*inv_norm, n_correct, nseqs, ntokens]; nll / ctc /
 * n_correct may be NULL; norm_out (optional f32) receives total*inv_norm, the value _train_step returns. */
int js2t_train_stats(double* stats6, const float* total, const float* nll, const float* ctc, const int64_t* n_correct,
                     double inv_norm, double nseqs, double ntokens, float* norm_out, js2t_stream stream);

/* out[0] = sum_i x[i] (single block, fixed order => bit-reproducible). */
int js2t_sum_f32(const float* x, int64_t n, float* out, js2t_stream stream);

/* nn.CTCLoss(blank, reduction='sum', zero_infinity) of loss.py:128-130,156-161 on LOGITS [B,T,V]
 * (batch-major; the reference's transpose(0,1) and log_softmax are folded in through `lse` = js2t_row_lse).
 * targets int64[B,Lmax] (padded), in_len/tgt_len int64[B].  alpha: f32[B,T,2*Lmax+1] (kept for backward),
 * nll: f32[B] raw negative log-likelihoods (inf when infeasible), loss_rows: f32[B] after zero_infinity.
 * beta (optional, f32[B,T,2*Lmax+1]): when given, the backward recursion runs in the same launch (it does not depend on
 * alpha) so that js2t_ctc_bwd(beta_ready = 1) only has the gradient pass left. */
/* row_offsets (optional, device int32 [B]): the logits / lse / dlogits are PACKED rows (js2t_pack_rows: the CTC branch of a ragged
 * batch on its live positions) - utterance b's frames are rows row_offsets[b] .. + in_len[b]; alpha / beta keep their [B,T,S]
 * layout (row_offsets then has B + 1 entries: the last one = the first row nobody owns); js2t_ctc_bwd writes the rows of live frames
 * and zeroes rows row_offsets[B] .. packed_rows of dlogits.  NULL: [B,T,V] as above. */
int js2t_ctc_alpha(const void* logits, int dt, const float* lse, const int64_t* targets, const int64_t* in_len,
                   const int64_t* tgt_len, float* alpha, float* beta, float* nll, float* loss_rows, int64_t B, int64_t T,
                   int64_t V, int64_t Lmax, int64_t blank, int zero_infinity, const int32_t* row_offsets, js2t_stream stream);
/* beta recursion (skipped when beta_ready != 0) + dlogits[b,t,v] = scale*(*g_dev)*(softmax_t(v) -
 * sum_{s:ext(s)=v} exp(alpha+beta-lp+nll)), zero for t >= in_len[b] and for infeasible utterances when zero_infinity.
 * beta: f32[B,T,2*Lmax+1] workspace, or the result of js2t_ctc_alpha. */
int js2t_ctc_bwd(const void* logits, int dt, const float* lse, const int64_t* targets, const int64_t* in_len,
                 const int64_t* tgt_len, const float* alpha, float* beta, const float* nll, const float* g_dev,
                 float scale, void* dlogits, int64_t B, int64_t T, int64_t V, int64_t Lmax, int64_t blank,
                 int zero_infinity, int beta_ready, const int32_t* row_offsets, int64_t packed_rows, js2t_stream stream);

/* Single-query attention for KV-cached decoding (replaces the per-step full-prefix decoder pass of search.py:518-534 and
 * the per-step re-projection of the encoder states, transformer_layers.py:75-107 under beam search):
 * out[r, h*dh:(h+1)*dh] = softmax_j( (q[r,h]/sqrt(dh)) . K[row(r,j), j, h] ) V[row(r,j), j, h],  j < len.
 * k / v: [*, Tmax, ldkv] (v may point into the same buffer as k).  idx_ld > 0: self-attention cache, row(r,j) =
 * idx[r*idx_ld + j] (per-position ancestry table: beam re-ordering rewrites the table, never the cache);
 * idx_ld == 0: cross-attention, row(r,j) = idx[r] (the hypothesis' utterance).  key_mask (optional) u8[*, Tmax] indexed
 * like k: 0 = masked.  `scale` multiplies q before the product (pass 1/sqrt(dh)).  len_dev (optional, device i32): the
 * key count is read from device memory instead of `len`, so one captured hipGraph of a decoding step can be replayed
 * for every position.  group > 1 (cross-attention only): hypotheses r, r+1, .. r+group-1 (r a multiple of group) share
 * idx[r] - the beams of one utterance - and are served by one block, which reads the keys / values once for all of them. */
int js2t_attn_decode(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, const int32_t* idx, int32_t idx_ld,
                     int32_t Tmax, int32_t len, const int32_t* len_dev, const uint8_t* key_mask, void* out, int64_t ldo, int32_t rows,
                     int32_t H, int32_t dh, float scale, int32_t group, int dt, js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * Conformer convolution module (reference transformer_layers.py:410-475, ConformerEncoderLayer :478-565).
 * The reference hands the module x.transpose(0, 1) of a [B, T, C] tensor (:549-552): its depthwise Conv1d and the
 * BatchNorm1d "length" axis therefore run over the BATCH index.  The kernels work on the [B, T, C] row-major activations
 * directly: js2t_dwconv_outer convolves along the OUTER index of an [L, N, C] array.
 * -------------------------------------------------------------------------------------------------- */
/* y[l,n,c] = bias[c] + sum_k w[c,k] * x[l + k - (K-1)/2, n, c]  (zero outside [0, L)); w: f32[C,K] (nn.Conv1d weight
 * [C,1,K], groups = C), bias: f32[C]; x, y: dt [L,N,C]. */
int js2t_dwconv_outer_fwd(const void* x, const float* w, const float* bias, void* y, int64_t L, int64_t N, int64_t C, int K, int dt,
                          js2t_stream stream);
/* dx (may be NULL) = correlation of dy with the flipped taps; dw[c,k] += sum_{l,n} dy[l,n,c] x[l+k-pad,n,c] (f32 atomics onto
 * the existing contents; NULL to skip).  The bias gradient is a column sum of dy (js2t_colsum). */
int js2t_dwconv_outer_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, int64_t L, int64_t N, int64_t C, int K,
                          int dt, js2t_stream stream);
/* nn.BatchNorm1d(C) over the rows of x[rows, C] followed by an activation (the module applies nn.Hardswish):
 * train != 0: batch mean / biased variance -> mean[C], invstd[C] (saved for backward) and the running statistics are
 * updated in place with `momentum` (unbiased variance), as torch does; train == 0: the running statistics are used.
 * y = act((x - mean) * invstd * gamma + beta).  ws: f32[2*C] workspace. */
int js2t_bn_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* mean,
                    float* invstd, void* y, float* ws, int64_t rows, int64_t C, float eps, float momentum, int train, int act, int dt,
                    js2t_stream stream);
/* Backward of the above given the saved mean / invstd: dz = dy * act'(z); train != 0: dx = gamma*invstd*(dz - mean_r(dz) -
 * xhat*mean_r(dz*xhat)), else dx = gamma*invstd*dz; dgamma[c] += sum dz*xhat, dbeta[c] += sum dz (accumulated in place).
 * ws: f32[2*C] workspace. */
int js2t_bn_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* invstd,
                    void* dx, float* dgamma, float* dbeta, float* ws, int64_t rows, int64_t C, int train, int act, int dt,
                    js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * Audio front-end (raw waveform -> padded, normalised, augmented feature batch).
 */

/* Kaldi-compatible log-mel filterbank for a ragged batch of utterances — replaces
 * torchaudio.compliance.kaldi.fbank(waveform * 2**15, num_mel_bins, sample_frequency) as called at
 * helpers_for_audio.py:30-37,54 (snip_edges framing, DC removal, pre-emphasis, window, |FFT|^2, mel, log).
 *   wave        f32 samples, utterance u starts at wave[sample_off[u]]
 *   frame_off   int64[U+1] prefix sums of frames per utterance (T_u = 1 + (N_u - win_len)/shift); out is
 *               f32[frame_off[U], n_mel], utterance u occupying rows frame_off[u] .. frame_off[u+1]-1
 *   window      f32[win_len] (Povey);  tw_re/tw_im f32[n_fft/2] = cos/-sin(2 pi k / n_fft)
 *   mel bin m   = sum_{k < mel_len[m]} power[mel_start[m]+k] * mel_w[mel_woff[m]+k]; the weights of bin m + 1 follow those
 *               of bin m in mel_w (mel_woff[n_mel-1] + mel_len[n_mel-1] = number of weights)
 *   scale       multiplies the samples first (2**15);  log_floor = FLT_EPSILON. */
int js2t_fbank(const float* wave, const int64_t* sample_off, const int64_t* frame_off, int32_t U,
               int64_t total_frames, const float* window, const float* tw_re, const float* tw_im,
               const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_woff, const float* mel_w,
               float* out, int32_t win_len, int32_t shift, int32_t n_fft, int32_t n_mel, float scale, float preemph,
               float log_floor, js2t_stream stream);

/* Utterance-level CMVN statistics — CMVN.__call__ (data_augmentation.py:96-109): mean[u,c], istd[u,c] =
 * 1/sqrt(max(E[x^2] - mean^2, 1e-10)) and fill[u] = mean of the normalised spectrogram, which is the value
 * SpecAugment writes into its masks (data_augmentation.py:45-46).  f32[U,F], f32[U,F], f32[U].
 * max_frames > 0: only the first max_frames frames of an utterance count - SpeechProcessor.__call__ truncates an
 * over-long evaluation utterance BEFORE CMVN (tokenizers.py:474-487). */
int js2t_cmvn_stats(const float* feat, const int64_t* frame_off, int32_t U, int32_t F, float* mean, float* istd,
                    float* fill, int32_t norm_means, int32_t norm_vars, int64_t max_frames, js2t_stream stream);
/* The same statistics by EIGHT blocks per utterance (a 32-utterance batch otherwise keeps 32 of 256 CUs busy: 39 us) through a
 * caller-owned workspace of js2t_cmvn_stats_workspace(U, F) doubles; fixed summation orders, bit-reproducible. */
int64_t js2t_cmvn_stats_workspace(int32_t U, int32_t F);
int js2t_cmvn_stats_ws(const float* feat, const int64_t* frame_off, int32_t U, int32_t F, float* mean, float* istd, float* fill,
                       int32_t norm_means, int32_t norm_vars, int64_t max_frames, double* workspace, js2t_stream stream);

/* out[u,t,c] = masked((feat - mean) * istd) for t < T_u, pad_value beyond — CMVN apply + SpecAugment masks
 * (data_augmentation.py:54-68; mask parameters drawn on the host to keep np.random parity) + pad_features
 * (helpers_for_audio.py:130-170, pads with 1.0).  masks: int32[U,8] = (f0,f, f0,f, t0,t, t0,t) or NULL;
 * mean/istd NULL = no CMVN.  out: [U,Tmax,F] in out_dt. */
int js2t_feature_finalize(const float* feat, const int64_t* frame_off, const float* mean, const float* istd,
                          const float* fill, const int32_t* masks, void* out, int out_dt, int64_t U, int64_t Tmax,
                          int32_t F, float pad_value, js2t_stream stream);
/* Same with Tmax a bucket length >= the longest utterance: positions t >= *crop_t (device scalar = frames of the longest
 * utterance) are 0 for every utterance, as if the batch had been cropped there (encoders.py:356-359) and zero-padded. */
int js2t_feature_finalize_crop(const float* feat, const int64_t* frame_off, const float* mean, const float* istd,
                               const float* fill, const int32_t* masks, void* out, int out_dt, int64_t U, int64_t Tmax,
                               int32_t F, float pad_value, const int64_t* crop_t, js2t_stream stream);

/* CMVN and / or SpecAugment applied to the ragged features IN PLACE - the orders and mask counts the fused pair above does not
 * cover (tokenizers.py:480-492 with CMVN(before=False); data_augmentation.py:54-68 with freq_mask_n / time_mask_n > 2):
 * x = (x - mean[u,c]) * istd[u,c] when mean is given, then fill[u] inside any mask of utterance u.
 * masks: int32[U, n_freq + n_time, 2] = (start, width), the n_freq frequency masks first; NULL = none. */
int js2t_feature_transform(float* feat, const int64_t* frame_off, int32_t U, int32_t F, const float* mean, const float* istd,
                           const float* fill, const int32_t* masks, int32_t n_freq, int32_t n_time, js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * Deterministic mode (off by default).  The reference's set_seed (helpers.py:93-104) sets torch.backends.cudnn.deterministic:
 * two runs from one seed agree bit for bit.  With the switch on, every kernel of the Transformer S2T train step that otherwise
 * sums through floating-point atomics takes an ordered form:
 *   js2t_gemm / js2t_gemm_grouped: split_k > 1 runs un-split and ADDS onto C (what the atomics do) - fewer blocks, slower;
 *   js2t_layernorm_bwd*: accumulate mode goes through the partial slab + the fixed-order final kernel instead of atomics;
 *   js2t_embed_bwd: the first position of a token id owns the table row and adds the later ones in position order;
 *   js2t_ctc_bwd: the first occurrence of a label in the extended target sums its later occurrences in order.
 * Round 5 - the extension kernels (BASELINE config 5) as well:
 *   js2t_flash_attn_bwd / js2t_rel_bias_grad: the relative-position bias gradient is collected as a histogram of 2^-32 fixed-point
 *     integers (integer atomics commute) in a scratch buffer the library owns, then converted and added into d_rel_bias;
 *   js2t_dwconv_outer_bwd (weight gradient), js2t_bn_act_fwd / _bwd (batch statistics, parameter-gradient sums): ONE block per
 *     64 columns walks all rows, so every output receives a single sum formed in a fixed order - slower, deterministic.
 * The first call in this mode allocates the scratch (never inside a hipGraph capture).
 *
 * Round 6 - the mode belongs to a CALLER, not to the process (a thin, stateless boundary).  js2t_ctx: a handful of
 * settings a caller binds to ITS THREAD around its launches (js2t_ctx_bind returns what was bound before, to be put back; NULL =
 * nothing bound); every entry point reads the settings of the launch it makes from there.  Keys: JS2T_CTX_DETERMINISTIC (1 = the
 * ordered forms above) and the kernel-selection rules of js2t_gemm / js2t_gemm_grouped, which never change a result
 * (JS2T_CTX_GEMM_P192_MODE / _P192_RING / _WG256_MODE / _PANEL_MODE: the values of the setters of the same names; -1 = the automatic
 * rule).  Two train steps in one process - one of them deterministic - no longer share a switch (tests/test_abi.py).
 * js2t_set_deterministic and the js2t_gemm_*_mode setters remain as process-wide TEST overrides: set, they win over any context.
 * js2t_ctx_effective(key): the value the calling thread's next launch would see. */
typedef struct js2t_ctx_s* js2t_ctx;
enum {
  JS2T_CTX_DETERMINISTIC = 0,
  JS2T_CTX_GEMM_P192_MODE = 1,
  JS2T_CTX_GEMM_P192_RING = 2,
  JS2T_CTX_GEMM_WG256_MODE = 3,
  JS2T_CTX_GEMM_PANEL_MODE = 4,
  JS2T_CTX_NKEYS = 5
};
js2t_ctx js2t_ctx_create(void);
void js2t_ctx_destroy(js2t_ctx ctx);
int js2t_ctx_set(js2t_ctx ctx, int32_t key, int32_t value);
int js2t_ctx_get(js2t_ctx ctx, int32_t key);
js2t_ctx js2t_ctx_bind(js2t_ctx ctx);
int js2t_ctx_effective(int32_t key);
void js2t_set_deterministic(int on);
int js2t_get_deterministic(void);

/* --------------------------------------------------------------------------------------------------
 * Update tail over the flat parameter store (training.py:436-456).
 */

/* out2[0] = ||g||_2 over the whole flat gradient, out2[1] = min(1, max_norm/(norm + 1e-6)) — the coefficient
 * nn.utils.clip_grad_norm_ applies (builders.py:68-71; max_norm <= 0 disables clipping).
 * partial: f32[js2t_sumsq_partials(n)] workspace. */
int64_t js2t_sumsq_partials(int64_t n);
int js2t_grad_norm_clip(const float* g, int64_t n, float max_norm, float* partial, float* out2, js2t_stream stream);
/* The same in two steps, for a gradient whose norm is collected piecewise: js2t_sumsq_ranges adds the partial sums of squares
 * of n_ranges pieces of g (table int64[n_ranges, 3] on the device: {first element, elements, first partial}; a piece of n
 * elements takes js2t_sumsq_partials(n) partials) to what other kernels left in `partial` (js2t_gemm_desc.sumsq_partial);
 * js2t_norm_clip turns partial[0 .. n_partial) into out2 = {norm, clip coefficient}. */
int js2t_sumsq_ranges(const float* g, const int64_t* table, int32_t n_ranges, int64_t n_blocks, float* partial, js2t_stream stream);
int js2t_norm_clip(const float* partial, int64_t n_partial, float max_norm, float* out2, js2t_stream stream);

/* torch.optim.AdamW step (builders.py:112-114) over flat fp32 buffers, gradient pre-scaled by
 * gscale * (*gscale_dev) (clip coefficient / loss-scale), optional bf16 shadow write and gradient clear.
 * step is the 1-based update count used for bias correction.  lr_dev / step_dev (optional device scalars)
 * override lr / step so that a captured hipGraph can be replayed with a moving schedule. */
int js2t_adamw(float* p, float* g, float* exp_avg, float* exp_avg_sq, void* lp_bf16, int64_t n, float lr,
               float beta1, float beta2, float eps, float weight_decay, int64_t step, const float* gscale_dev,
               float gscale, int zero_grad, const float* lr_dev, const int64_t* step_dev, js2t_stream stream);

/* The same update over a table of pieces of the flat store, fused with what is derived from the new weights (the two launches
 * that used to follow every update: js2t_transpose_groups, js2t_fold_ln_weights):
 *   items int64[n_items, 8] = {kind, off, rows, cols, first unit, fold row | -1, keep gradient, 0}, units numbered
 *   consecutively; "keep gradient" != 0: this piece's gradient is NOT cleared even with zero_grad (its producer overwrites it);
 *   kind 0: elements [off, off + rows) (off, rows multiples of 4), ceil(rows / flat_unit) units;
 *   kind 1: a row-major fp32 [rows, cols] matrix at element `off` (off % 4 == 0, cols % 4 == 0),
 *           ceil(rows / unit_rows) * ceil(cols / unit_cols) units; with lp_t_bf16 its transposed bf16 image is written at
 *           lp_t_bf16 + off as [cols, rows]; fold (cols <= unit_cols only) names a row of `folds`, the table of
 *           js2t_fold_ln_weights, whose W is this matrix: Wf and bias_f of that row are written from the NEW weights and the
 *           CURRENT gamma / beta / bias (update those in an earlier launch).
 * Replaces, per optimizer update, torch.optim.AdamW.step (builders.py:112-114) + the re-derivation of every compute-dtype
 * copy of the weights.  js2t_adamw_items_geometry reports {flat_unit, unit_rows, unit_cols}. */
int js2t_adamw_items(float* p, float* g, float* exp_avg, float* exp_avg_sq, void* lp_bf16, void* lp_t_bf16,
                     const int64_t* items, int32_t n_items, int64_t n_units, const int64_t* folds, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int64_t step, const float* gscale_dev, float gscale,
                     int zero_grad, const float* lr_dev, const int64_t* step_dev, js2t_stream stream);
void js2t_adamw_items_geometry(int32_t* flat_unit, int32_t* rows, int32_t* cols);

/* --------------------------------------------------------------------------------------------------
 * Beam-search step (search.py:562-646): per live batch element, over its `beam` hypotheses:
 *   log_softmax(logits[row,:]) ; forbidden ids := -inf (search.py:590-601; forbid_ids is a HOST array) ;
 *   += beam_log_probs[row] (:622) ; *= 1/length_penalty when length_penalty > 0 (:626-628) ;
 *   top-`beam` of the beam*V flattened scores (:632-636), ordered by score descending then flat index ascending.
 * logits f32[n_batch*beam, V]; beam_log_probs f32[n_batch*beam]; out_scores f32[n_batch, beam] (penalised);
 * out_ids int64[n_batch, beam] (flat index = beam_index*V + token); out_lse f32[n_batch*beam] (row log-sum-exp). */
int js2t_beam_step(const float* logits, const float* beam_log_probs, float* out_scores, int64_t* out_ids,
                   float* out_lse, int64_t n_batch, int32_t beam, int64_t V, const int32_t* forbid_ids,
                   int32_t n_forbid, float length_penalty, js2t_stream stream);

/* The same step over rows that already ARE log-probabilities (no normalisation; out_lse := 0): the decoding options
 * of search.py:564-618 edit the log-softmax output before the beam scores are added - n-gram blocking, repetition
 * penalty, forced prompt tokens - so on that path log_softmax (js2t_log_softmax), the edits below and this selection
 * are separate launches. */
int js2t_beam_step_logp(const float* log_probs, const float* beam_log_probs, float* out_scores, int64_t* out_ids,
                        float* out_lse, int64_t n_batch, int32_t beam, int64_t V, const int32_t* forbid_ids,
                        int32_t n_forbid, float length_penalty, js2t_stream stream);

/* EXTENSION (the reference returns ctc_out for return_type="decode_ctc", joeynmt/model.py:162-166, and has no
 * consumer): joint CTC / attention decoding after Watanabe et al., IEEE JSTSP 2017, Algorithm 2.
 * js2t_beam_pick: for every row of logits [rows, V] the n_pick (<= 8) best tokens by log-softmax (forbidden ids masked) +
 * row_scores[row]: out_scores / out_ids [rows, n_pick] (score descending, id ascending among ties), out_lse [rows].
 * js2t_ctc_prefix_step: one extension step of the CTC prefix score for every (hypothesis row, candidate) pair.  ctc_log_probs f32
 * [B, T, V] (B = rows / beam utterances), in_len [B]; r_prev f32 [rows, T, 2] forward variables (non-blank, blank) of the
 * hypotheses (log 0 = -1e30), last_tok [rows] their last tokens, n_out = tokens emitted so far (0: only BOS); cand / cand_lp
 * [rows, n_cand] candidate ids and attention log-probabilities; psi_prev [rows] prefix scores of the hypotheses.  Writes psi_out
 * [rows, n_cand], r_new [rows, n_cand, T, 2] and local [rows, n_cand] = (1 - weight) * cand_lp + weight * (psi - psi_prev)
 * (-inf where either probability is 0) - the step score js2t_beam_step_logp then selects on. */
int js2t_beam_pick(const float* logits, const float* row_scores, float* out_scores, int64_t* out_ids, float* out_lse, int64_t rows,
                   int32_t n_pick, int64_t V, const int32_t* forbid_ids, int32_t n_forbid, js2t_stream stream);
int js2t_ctc_prefix_step(const float* ctc_log_probs, const int64_t* in_len, const float* r_prev, const int64_t* last_tok,
                         const int64_t* cand, const float* cand_lp, const float* psi_prev, float* local, float* psi_out, float* r_new,
                         int64_t rows, int32_t beam, int32_t n_cand, int32_t T, int64_t V, int32_t n_out, int32_t blank, int32_t eos,
                         float weight, js2t_stream stream);
/* Test hook: 1 = the thread-per-pair form of js2t_ctc_prefix_step (round 5; also what runs when an utterance's frames do not fit
 * 64 KB of LDS), 0 = the default, a block per hypothesis with the operands staged through LDS.  Bit-identical results. */
void js2t_debug_ctc_prefix_thread_per_pair(int on);

/* penalize_repetition (search.py:972-1001): for every id in tokens[row, 0..L) (int64[rows, L]; hypothesis prefix or
 * source tokens) log_probs[row, id] := x * penalty if x < 0 else x / penalty, x = the value BEFORE this call (gather,
 * scale, scatter: an id that occurs several times is penalised once).  In place on f32[rows, V].
 * The reference's `exclude_tokens` restore is a no-op there (scores_before aliases scores, :986,996-999): none here. */
int js2t_rep_penalty(float* log_probs, const int64_t* tokens, int64_t rows, int64_t V, int64_t L, float penalty,
                     js2t_stream stream);

/* log_probs[rows[i], cols[i]] := value for i < n (device index arrays): the -inf of block_repeat_ngrams
 * (search.py:966-969) and of the forbidden ids (:590-601), the 0 of a forced prompt token (:614-618). */
int js2t_logp_set(float* log_probs, const int64_t* rows, const int64_t* cols, int64_t n, int64_t V, float value,
                  js2t_stream stream);

/* --------------------------------------------------------------------------------------------------
 * Fused multi-head attention (bf16, head size 128 or 64): softmax(mask(q k^T * scale [+ rel]))) [dropout] v without
 * materialising the [B,H,Tq,Tk] scores — MultiHeadedAttention.forward, transformer_layers.py:86-105, and its
 * backward.  Head h of token (b,t) lives at ptr[(b*T + t)*ld + h*128 ..]; pointers are pre-offset into fused
 * k|v|q buffers.  mask: uint8, element (b,q,k) at mask[b*mask_sb + q*mask_sq + k] (mask_sq = 0 for key padding),
 * NULL = none.  lse: f32[B*H, Tq] row log-sum-exp (written by fwd, read by bwd); delta: f32[B*H, Tq] workspace.
 * Dropout draws the same masks as js2t_softmax_fwd for equal (rng_state, rng_stream).
 * rel_bias (extension for BASELINE config 5, "rel-pos attn"; the reference's attention has no relative term,
 * transformer_layers.py:49-115): f32[H, 2*rel_R + 1], added to the scaled score of (query i, key j) of head h as
 * rel_bias[h][clamp(j - i, -rel_R, rel_R) + rel_R] (a learned bias per head and clipped distance); d_rel_bias, same
 * shape, receives the gradient by f32 atomics (+=: zero it first).  NULL = off.
 */
typedef struct js2t_attn_desc {
  const void* q; const void* k; const void* v;
  void* o;            /* fwd: output [B*Tq, ldo]; bwd: the forward output (input) */
  const void* d_o;    /* bwd: gradient of o */
  void* dq; void* dk; void* dv; /* bwd outputs */
  float* lse; float* delta;
  const uint8_t* mask;
  int64_t ldq, ldk, ldv, ldo, ld_do, ld_dq, ld_dk, ld_dv, mask_sb, mask_sq;
  int32_t B, H, Tq, Tk, head_dim;
  float scale, dropout_p;
  const uint64_t* rng_state;
  uint32_t rng_stream;
  int32_t rel_R;            /* clipping distance of the relative-position bias, 1..255 */
  const float* rel_bias;    /* f32[H, 2 rel_R + 1] or NULL */
  float* d_rel_bias;        /* bwd: += gradient of rel_bias (NULL: not wanted) */
  /* bwd, optional: delta = rowsum(dO * O) arrives as partial sums over 64-column groups of the [B*Tq, H*head_dim] layout
   * (js2t_gemm_desc.dot_partial of the product that made dO): f32 [B*Tq, delta_groups], head h owning groups h * head_dim / 64 ..;
   * `o` and `delta` are then not read / written and the dQ and dK/dV passes run as ONE grid (two launches otherwise: the
   * dK/dV pass waits for the delta the dQ pass computes). */
  const float* delta_partial;
  int32_t delta_groups;
  /* optional, self-attention over PACKED rows (the encoder of a ragged batch, encoders.py:348-373 of the reference: positions behind
   * an utterance's length are dead): device int32 [B + 1]; batch entry b owns rows seg[b] .. seg[b+1] of q / k / v / o / d_o /
   * dq / dk / dv and has seg[b+1] - seg[b] <= Tq positions.  Tq == Tk is then the longest entry: it shapes the grid, lse / delta
   * ([B*H, Tq]) and the mask rows, and keeps the dropout counters those of the padded layout.  Tiles behind an entry's length
   * are not touched.  seg_rows = rows of the packed buffers (>= seg[B]; callers round it up to a bucket): rows seg[B] .. seg_rows of
   * o (forward) / dq, dk, dv (backward), which no entry owns, are ZEROED by the kernels - they are read by the products behind
   * (whose column sums - weight gradients, bias gradients - run over all rows); 0: not written, the caller's business. */
  const int32_t* seg;
  int64_t seg_rows;
  /* seg_keys != 0: CROSS-attention over packed keys (the decoder reading the encoder states of a ragged batch, transformer_layers.py:
   * 383-396 of the reference): seg describes k / v / dk / dv only - entry b's keys are rows seg[b] .. seg[b+1] and it has
   * seg[b+1] - seg[b] <= Tk of them; q / o / d_o / dq stay in the padded [B * Tq] layout.  Tk is the longest entry (grid shape, mask
   * rows, dropout counters of the padded layout); seg_rows zeroes the tails of dk / dv. */
  int32_t seg_keys;
} js2t_attn_desc;
int js2t_flash_attn_fwd(const js2t_attn_desc* d, js2t_stream stream);
int js2t_flash_attn_bwd(const js2t_attn_desc* d, js2t_stream stream);
/* Measurement switches (tools/, tests): js2t_debug_attn_fwd_sb(0 / 1 / -1) forces the double- / single-buffered forward kernel or
 * returns to the per-launch rule; js2t_debug_attn_bwd_merge(0) runs the backward as two launches also when delta_partial is given
 * (1 = one grid, the default).  Process-wide, read at launch time; results do not depend on them. */
void js2t_debug_attn_fwd_sb(int mode);
void js2t_debug_attn_bwd_merge(int on);

/* --------------------------------------------------------------------------------------------------
 * Gradient exchange (data parallel, one process per GPU): what DistributedDataParallel's bucketed all-reduce does for the
 * reference (prediction.py:508-515 wraps the model; helpers_for_ddp.py:17-38 sets the process group up, :157-174 reduces
 * scalars) - an RCCL communicator over xGMI.  Bootstrap: rank 0 calls js2t_comm_unique_id, the host side broadcasts the
 * js2t_comm_unique_id_bytes() bytes by whatever channel it has (torch.distributed / gloo store), every rank calls
 * js2t_comm_init with them (collective: returns when all `world` ranks have).  librccl is resolved at first use.
 *
 * js2t_comm_allreduce_async: in-place sum (average != 0: mean) of buf[count] (JS2T_F32 or JS2T_BF16) over all ranks, enqueued
 * on the communicator's own stream behind everything `producer` has been given so far; returns at once.  Calls on one
 * communicator run in call order; every rank must issue the same sequence.
 * js2t_comm_wait: `consumer` (a stream) waits on the device for everything enqueued so far (host == 0), or the calling
 * thread does (host != 0; consumer ignored).  js2t_comm_stream: the communicator's stream, for work that belongs between
 * two collectives (casts into / out of a bf16 staging buffer).  js2t_comm_destroy drains it first.
 */
int64_t js2t_comm_unique_id_bytes(void);
int js2t_comm_unique_id(void* out, int64_t nbytes);
int js2t_comm_init(void** comm_out, const void* unique_id, int64_t id_bytes, int32_t world, int32_t rank, int32_t device);
int js2t_comm_allreduce_async(void* comm, void* buf, int64_t count, int32_t dtype, int32_t average, js2t_stream producer);
int js2t_comm_wait(void* comm, js2t_stream consumer, int32_t host);
js2t_stream js2t_comm_stream(void* comm);
int js2t_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* JOEYS2T_HIP_H */
