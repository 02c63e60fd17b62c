// Audio front-end kernels: Kaldi-compatible log-mel filterbank (one wavefront per frame, 512-point FFT in LDS),
// utterance-level CMVN statistics, and the fused normalise + SpecAugment + pad-to-batch writer.
// HBM-bound: ~0.96 MB of waveform read and ~0.48 MB of features written per 15 s utterance.
//
// Replaces torchaudio.compliance.kaldi.fbank as called at helpers_for_audio.py:30-37,54 (Kaldi defaults: snip
// edges, no dither, DC removal, pre-emphasis 0.97, Povey window, power spectrum, log with floor eps), CMVN
// (data_augmentation.py:96-109), SpecAugment's masking (data_augmentation.py:54-68) and pad_features
// (helpers_for_audio.py:130-170).
#include "common.hpp"

namespace {

constexpr int FB_WAVES = 4;       // frames per 256-thread block
constexpr int FB_MAX_FFT = 1024;  // complex points per frame held in LDS

// utterance of a global frame index: largest u with frame_off[u] <= f
__device__ __forceinline__ int find_utt(const int64_t* __restrict__ frame_off, int U, int64_t f) {
  int lo = 0, hi = U;  // invariant frame_off[lo] <= f < frame_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (frame_off[mid] <= f) lo = mid; else hi = mid;
  }
  return lo;
}

// One wave = one frame.  wave: f32 samples in [-1,1]; sample_off[u] = start of utterance u in `wave`.
__global__ __launch_bounds__(256) void fbank_kernel(const float* __restrict__ wave, const int64_t* __restrict__ sample_off,
                                                    const int64_t* __restrict__ frame_off, int U,
                                                    const float* __restrict__ window, const float* __restrict__ tw_re,
                                                    const float* __restrict__ tw_im, const int32_t* __restrict__ mel_start,
                                                    const int32_t* __restrict__ mel_len, const int32_t* __restrict__ mel_woff,
                                                    const float* __restrict__ mel_w, float* __restrict__ out, int win_len,
                                                    int shift, int n_fft, int log2_fft, int n_mel, float scale,
                                                    float preemph, float log_floor) {
  __shared__ float sre[FB_WAVES][FB_MAX_FFT];
  __shared__ float sim[FB_WAVES][FB_MAX_FFT];
  __shared__ float stw_re[FB_MAX_FFT / 2], stw_im[FB_MAX_FFT / 2];  // twiddles: read every stage, fetched from HBM once per block
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (n_fft >> 1); i += 256) {
    stw_re[i] = tw_re[i];
    stw_im[i] = tw_im[i];
  }
  const int64_t total = frame_off[U];
  const int64_t f = (int64_t)blockIdx.x * FB_WAVES + w;
  const bool live = f < total;  // keep all waves in the barriers
  float* re = sre[w];
  float* im = sim[w];
  int u = 0;
  int64_t t = 0;
  if (live) {
    u = find_utt(frame_off, U, f);
    t = f - frame_off[u];
  }
  // 1. load + scale, DC removal
  const float* x = wave + (live ? sample_off[u] + t * shift : 0);
  float s = 0.f;
  for (int i = lane; i < win_len; i += 64) {
    const float v = live ? x[i] * scale : 0.f;
    re[i] = v;
    s += v;
  }
  const float mean = wave_sum(s) / (float)win_len;
  __syncthreads();
  // 2. pre-emphasis (x[-1] := x[0]) + window, written in bit-reversed order for the in-place DIT FFT
  float fr[FB_MAX_FFT / 64];
#pragma unroll
  for (int j = 0; j < FB_MAX_FFT / 64; ++j) {
    const int i = lane + 64 * j;
    float v = 0.f;
    if (i < win_len) {
      const float cur = re[i] - mean;
      const float prev = re[i > 0 ? i - 1 : 0] - mean;
      v = (cur - preemph * prev) * window[i];
    }
    fr[j] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < FB_MAX_FFT / 64; ++j) {
    const int i = lane + 64 * j;
    if (i < n_fft) {
      const int r = (int)(__brev((unsigned)i) >> (32 - log2_fft));
      re[r] = fr[j];
      im[r] = 0.f;
    }
  }
  __syncthreads();
  // 3. radix-2 decimation-in-time FFT, n_fft/2 butterflies per stage spread over the 64 lanes
  for (int sgs = 1; sgs <= log2_fft; ++sgs) {
    const int half = 1 << (sgs - 1);
    const int tw_step = n_fft >> sgs;
    for (int b = lane; b < (n_fft >> 1); b += 64) {
      const int k = b & (half - 1);
      const int i0 = ((b >> (sgs - 1)) << sgs) + k, i1 = i0 + half;
      const float wr = stw_re[k * tw_step], wi = stw_im[k * tw_step];
      const float xr = re[i1], xi = im[i1];
      const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
      const float ar = re[i0], ai = im[i0];
      re[i1] = ar - tr; im[i1] = ai - ti;
      re[i0] = ar + tr; im[i0] = ai + ti;
    }
    // a frame belongs to one wave: its LDS accesses execute in issue order, so the stages only need the compiler (and the
    // counter) kept honest - no block barrier between them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // 4. power spectrum of bins 0 .. n_fft/2 - 1 (the Nyquist bin has zero mel weight)
  for (int k = lane; k < (n_fft >> 1); k += 64) {
    const float a = re[k], b = im[k];
    im[k + (n_fft >> 1)] = a * a + b * b;  // stash in the upper half of `im`
  }
  __syncthreads();
  // 5. mel bins + log
  if (live) {
    const float* pw = im + (n_fft >> 1);
    for (int m = lane; m < n_mel; m += 64) {
      const int st = mel_start[m], ln = mel_len[m];
      const float* wt = mel_w + mel_woff[m];
      float e = 0.f;
      for (int k = 0; k < ln; ++k) e += pw[st + k] * wt[k];
      out[f * n_mel + m] = __logf(fmaxf(e, log_floor));
    }
  }
}


// ---------------------------------------------------------------- 512-point frames: the FFT lives in registers
// The kernel above keeps a frame's 512 complex points in LDS and makes nine radix-2 passes over them: ~500 LDS
// wave-instructions per frame, most of them 2- to 8-way bank-conflicted (strided butterflies, strided twiddles) - the CU's
// LDS pipe is what bounds it (228 us for 47,936 frames, 2 % of the HBM rate it is nominally bound by).  Here a lane holds 8
// points; a decimation-in-frequency FFT does three radix-2 stages in registers, transposes through LDS (conflict-free padded
// images), three more stages, a second transpose, the last three stages: 2 x 16 LDS stores + loads per lane instead of ~500,
// twiddles of the first six stages in registers for the whole kernel (they depend on the lane, not on the frame), those of
// the last three are constants.  A wave walks over frames with a grid stride.
//   position n = 64 j + l (lane l, register j)  --T1-->  n = 64 j' + 8 m + r (lane 8 j' + r, register m)  --T2-->  lane 8 j' + m,
//   register r.  DIF leaves X[bitrev9(n)] at position n: bins below Nyquist are the even r.
struct cplx { float re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// DIF butterfly: (a, b) -> (a + b, (a - b) * w)
__device__ __forceinline__ void bfly(cplx& a, cplx& b, cplx w) {
  const cplx d = {a.re - b.re, a.im - b.im};
  a = {a.re + b.re, a.im + b.im};
  b = cmul(d, w);
}
__device__ __forceinline__ void bfly1(cplx& a, cplx& b) {  // w = 1
  const cplx d = {a.re - b.re, a.im - b.im};
  a = {a.re + b.re, a.im + b.im};
  b = d;
}
__device__ __forceinline__ int rev3(int v) { return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1); }

constexpr int FB5_T1 = 72, FB5_T2 = 9;  // padded strides (float2 elements) of the two transpose images
constexpr int FB5_MELW = 2048;          // filter weights kept in LDS (80 bins over 256 FFT bins have ~1000)
__global__ __launch_bounds__(256) void fbank512_kernel(const float* __restrict__ wave, const int64_t* __restrict__ sample_off,
                                                       const int64_t* __restrict__ frame_off, int U,
                                                       const float* __restrict__ window, const float* __restrict__ tw_re,
                                                       const float* __restrict__ tw_im, const int32_t* __restrict__ mel_start,
                                                       const int32_t* __restrict__ mel_len, const int32_t* __restrict__ mel_woff,
                                                       const float* __restrict__ mel_w, float* __restrict__ out, int win_len,
                                                       int shift, int n_mel, float scale, float preemph, float log_floor) {
  __shared__ float2 simg[FB_WAVES][8 * FB5_T1];  // 576 float2 >= 64 * FB5_T2
  __shared__ float spw[FB_WAVES][256];
  __shared__ float smw[FB5_MELW];  // the sparse filter weights: read by every frame, fetched once per block
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float2* img = simg[w];
  float* pw = spw[w];
  const int n_mel_w = mel_woff[n_mel - 1] + mel_len[n_mel - 1];  // the filters' weights are stored back to back, in bin order
  const bool mw_lds = n_mel_w <= FB5_MELW;
  if (mw_lds)
    for (int i = threadIdx.x; i < n_mel_w; i += 256) smw[i] = mel_w[i];
  __syncthreads();
  // this lane's (at most two) mel bins: bin lane and bin lane + 64
  int mst[2], mln[2], mwo[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int mb = lane + 64 * q;
    mst[q] = mb < n_mel ? mel_start[mb] : 0;
    mln[q] = mb < n_mel ? mel_len[mb] : 0;
    mwo[q] = mb < n_mel ? mel_woff[mb] : 0;
  }
  auto tw = [&](int t) -> cplx { return {tw_re[t], tw_im[t]}; };  // exp(-2 pi i t / 512), t < 256
  // twiddles of stages 1-6 (W_512^t): they depend on the lane and the register index only
  const int r = lane & 7;
  cplx w1[4], w2[2], w3, w4[4], w5[2], w6;
#pragma unroll
  for (int j = 0; j < 4; ++j) w1[j] = tw(lane + 64 * j);
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = tw(2 * (lane + 64 * j));
  w3 = tw(4 * lane);
#pragma unroll
  for (int m = 0; m < 4; ++m) w4[m] = tw(8 * (r + 8 * m));
#pragma unroll
  for (int m = 0; m < 2; ++m) w5[m] = tw(16 * (r + 8 * m));
  w6 = tw(32 * r);
  const cplx c8[4] = {{1.f, 0.f}, {0.70710678118654752f, -0.70710678118654752f}, {0.f, -1.f}, {-0.70710678118654752f, -0.70710678118654752f}};
  float win[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) win[j] = lane + 64 * j < win_len ? window[lane + 64 * j] : 0.f;
  // a wave takes a run of consecutive frames: one utterance search per run, neighbouring frames share 60 % of their samples
  const int64_t total = frame_off[U];
  const int64_t n_waves = (int64_t)gridDim.x * FB_WAVES, per = (total + n_waves - 1) / n_waves;
  const int64_t f0 = ((int64_t)blockIdx.x * FB_WAVES + w) * per, f1 = min(total, f0 + per);
  int u = f0 < total ? find_utt(frame_off, U, f0) : 0;
  int64_t u_end = f0 < total ? frame_off[u + 1] : 0, u_beg = f0 < total ? frame_off[u] : 0, s_off = f0 < total ? sample_off[u] : 0;
  // samples of the NEXT frame are requested while the current one is transformed (three waves per SIMD do not cover an HBM
  // round trip on their own); a lane's predecessor sample sits in the neighbouring lane (DPP wave shift), lane 0's in lane
  // 63 of the previous register
  auto frame_ptr = [&](int64_t f) -> const float* {
    while (f >= u_end) {  // next utterance (empty ones are skipped)
      ++u;
      u_beg = u_end;
      u_end = frame_off[u + 1];
      s_off = sample_off[u];
    }
    return wave + s_off + (f - u_beg) * shift;
  };
  float nxt[8];
  if (f0 < f1) {
    const float* x0 = frame_ptr(f0);
#pragma unroll
    for (int j = 0; j < 8; ++j) nxt[j] = lane + 64 * j < win_len ? x0[lane + 64 * j] : 0.f;
  }
  for (int64_t f = f0; f < f1; ++f) {
    // 1. scale, DC removal, pre-emphasis (x[-1] := x[0]), window
    float cur[8], prv[8], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      cur[j] = nxt[j] * scale;
      sum += cur[j];
    }
    if (f + 1 < f1) {
      const float* xn = frame_ptr(f + 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) nxt[j] = lane + 64 * j < win_len ? xn[lane + 64 * j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cur[j]), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
      // lane 0: sample 64 j - 1 is lane 63's register j - 1; for the very first sample x[-1] := x[0] (bit patterns travel
      // through the scalar unit: the readlane builtins take integers)
      const int edge = j == 0 ? __builtin_amdgcn_readfirstlane(__float_as_int(cur[0]))
                              : __builtin_amdgcn_readlane(__float_as_int(cur[j > 0 ? j - 1 : 0]), 63);
      prv[j] = lane == 0 ? __int_as_float(edge) : up;
    }
    const float mean = wave_sum(sum) / (float)win_len;
    cplx a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = {lane + 64 * j < win_len ? ((cur[j] - mean) - preemph * (prv[j] - mean)) * win[j] : 0.f, 0.f};
    // 2. stages 1-3 (spans 256, 128, 64): register pairs (j, j+4), (j, j+2), (j, j+1)
#pragma unroll
    for (int j = 0; j < 4; ++j) bfly(a[j], a[j + 4], w1[j]);
#pragma unroll
    for (int hlf = 0; hlf < 8; hlf += 4)
#pragma unroll
      for (int j = 0; j < 2; ++j) bfly(a[hlf + j], a[hlf + j + 2], w2[j]);
#pragma unroll
    for (int j = 0; j < 8; j += 2) bfly(a[j], a[j + 1], w3);
    // 3. transpose 1: (j, l) -> lane 8 j' + r holds positions r + 8 m of sub-transform j'
#pragma unroll
    for (int j = 0; j < 8; ++j) img[j * FB5_T1 + lane] = make_float2(a[j].re, a[j].im);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int jp = lane >> 3;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float2 v = img[jp * FB5_T1 + r + 8 * m];
      a[m] = {v.x, v.y};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 4. stages 4-6 (spans 32, 16, 8 inside a 64-point sub-transform): register pairs (m, m+4), (m, m+2), (m, m+1)
#pragma unroll
    for (int m = 0; m < 4; ++m) bfly(a[m], a[m + 4], w4[m]);
#pragma unroll
    for (int hlf = 0; hlf < 8; hlf += 4)
#pragma unroll
      for (int m = 0; m < 2; ++m) bfly(a[hlf + m], a[hlf + m + 2], w5[m]);
#pragma unroll
    for (int m = 0; m < 8; m += 2) bfly(a[m], a[m + 1], w6);
    // 5. transpose 2: lane 8 j' + m gathers r = 0..7 of its 8-point block
#pragma unroll
    for (int m = 0; m < 8; ++m) img[(8 * jp + m) * FB5_T2 + r] = make_float2(a[m].re, a[m].im);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float2 v = img[lane * FB5_T2 + q];
      a[q] = {v.x, v.y};
    }
    // 6. stages 7-9 (spans 4, 2, 1): constants W_8^q, W_4^q
#pragma unroll
    for (int q = 0; q < 4; ++q) bfly(a[q], a[q + 4], c8[q]);
#pragma unroll
    for (int hlf = 0; hlf < 8; hlf += 4) {
      bfly1(a[hlf], a[hlf + 2]);
      bfly(a[hlf + 1], a[hlf + 3], c8[2]);
    }
#pragma unroll
    for (int q = 0; q < 8; q += 2) bfly1(a[q], a[q + 1]);
    // 7. power spectrum of the bins below Nyquist: position n = 64 j' + 8 m + q holds X[bitrev9(n)]; bit 8 of the bin is q & 1
    const int mm = lane & 7, base = 8 * rev3(mm) + rev3(jp);
#pragma unroll
    for (int q = 0; q < 8; q += 2) pw[64 * rev3(q) + base] = a[q].re * a[q].re + a[q].im * a[q].im;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 8. mel bins + log: bins lane and lane + 64 from registers-resident filter descriptors, weights in LDS; four partial
    //    sums per bin (the order of the additions inside a bin is fixed: results do not depend on the launch)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int mb = lane + 64 * q;
      if (mb < n_mel) {
        const float* wt = mw_lds ? smw + mwo[q] : mel_w + mwo[q];
        const float* pp = pw + mst[q];
        float e0 = 0.f, e1 = 0.f, e2 = 0.f, e3 = 0.f;
        int k = 0;
        for (; k + 4 <= mln[q]; k += 4) {
          e0 += pp[k] * wt[k];
          e1 += pp[k + 1] * wt[k + 1];
          e2 += pp[k + 2] * wt[k + 2];
          e3 += pp[k + 3] * wt[k + 3];
        }
        for (; k < mln[q]; ++k) e0 += pp[k] * wt[k];
        out[f * n_mel + mb] = __logf(fmaxf((e0 + e1) + (e2 + e3), log_floor));
      }
    }
    for (int mb = lane + 128; mb < n_mel; mb += 64) {  // more than 128 bins: the plain loop
      const int st = mel_start[mb], ln = mel_len[mb];
      const float* wt = mel_w + mel_woff[mb];
      float e = 0.f;
      for (int k = 0; k < ln; ++k) e += pw[st + k] * wt[k];
      out[f * n_mel + mb] = __logf(fmaxf(e, log_floor));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // pw / img are rewritten by the next frame
  }
}

// ---------------------------------------------------------------- CMVN statistics (one block per utterance)
// mean[u,c] = mean_t x ; istd[u,c] = 1/sqrt(max(sum x^2 / T - mean^2, 1e-10)) ; fill[u] = mean of the normalised
// spectrogram (SpecAugment's mask value).  Accumulated in f64.
constexpr int CM_GROUPS = 12;  // frame groups per block: 12 x 80 bins = 960 threads share a piece of an utterance
constexpr int CM_SPLIT = 8;    // pieces per utterance: 32 utterances alone would keep 32 of 256 CUs busy pulling 480 KB each (39 us)
// phase 1: block (u, p) sums piece p of utterance u's kept frames -> part[u][p][2][F] (f64); phase 2 (cmvn_finish_kernel): one block
// per utterance adds the pieces in order and writes mean / istd / fill.  Fixed orders: bit-reproducible.
__global__ void cmvn_partial_kernel(const float* __restrict__ feat, const int64_t* __restrict__ frame_off, int F, double* __restrict__ part,
                                    int64_t max_frames) {
  extern __shared__ double sh[];  // [2][CM_GROUPS][F]
  const int u = blockIdx.x, p = blockIdx.y;
  // an over-long evaluation utterance is cut to max_length BEFORE CMVN (tokenizers.py:474-487): statistics over the kept frames
  const int64_t t0 = frame_off[u], T = max_frames > 0 ? min(frame_off[u + 1] - t0, max_frames) : frame_off[u + 1] - t0;
  const int64_t per = (T + CM_SPLIT - 1) / CM_SPLIT, ta = min(T, p * per), tb = min(T, ta + per);
  const int c = threadIdx.x % F, gi = threadIdx.x / F;
  double s = 0.0, q = 0.0;
  if (gi < CM_GROUPS) {
    for (int64_t t = ta + gi; t < tb; t += CM_GROUPS) {
      const double v = feat[(t0 + t) * F + c];
      s += v;
      q += v * v;
    }
    sh[gi * F + c] = s;
    sh[(CM_GROUPS + gi) * F + c] = q;
  }
  __syncthreads();
  if (threadIdx.x < F) {
    double ss = 0.0, qq = 0.0;
    for (int g2 = 0; g2 < CM_GROUPS; ++g2) {
      ss += sh[g2 * F + c];
      qq += sh[(CM_GROUPS + g2) * F + c];
    }
    double* out = part + ((int64_t)u * CM_SPLIT + p) * 2 * F;
    out[c] = ss;
    out[F + c] = qq;
  }
}
__global__ void cmvn_finish_kernel(const double* __restrict__ part, const int64_t* __restrict__ frame_off, int F, float* __restrict__ mean,
                                   float* __restrict__ istd, float* __restrict__ fill, int norm_means, int norm_vars, int64_t max_frames) {
  extern __shared__ double norm_mean[];  // [F]
  const int u = blockIdx.x, c = threadIdx.x;
  const int64_t t0 = frame_off[u], T = max_frames > 0 ? min(frame_off[u + 1] - t0, max_frames) : frame_off[u + 1] - t0;
  if (c < F) {
    double ss = 0.0, qq = 0.0;
    for (int p = 0; p < CM_SPLIT; ++p) {
      const double* in = part + ((int64_t)u * CM_SPLIT + p) * 2 * F;
      ss += in[c];
      qq += in[F + c];
    }
    const float m = T > 0 ? (float)(ss / (double)T) : 0.f;
    const float sq = T > 0 ? (float)(qq / (double)T) : 0.f;
    const float var = sq - m * m;
    const float is = norm_vars ? 1.f / sqrtf(fmaxf(var, 1e-10f)) : 1.f;
    mean[(int64_t)u * F + c] = norm_means ? m : 0.f;
    istd[(int64_t)u * F + c] = is;
    norm_mean[c] = (double)((norm_means ? 0.f : m) * is);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int k = 0; k < F; ++k) a += norm_mean[k];
    fill[u] = (float)(a / (double)F);
  }
}

// the round-1 form: one block per utterance, no workspace (js2t_cmvn_stats)
__global__ void cmvn_stats_kernel(const float* __restrict__ feat, const int64_t* __restrict__ frame_off, int F,
                                  float* __restrict__ mean, float* __restrict__ istd, float* __restrict__ fill, int norm_means,
                                  int norm_vars, int64_t max_frames) {
  extern __shared__ double sh[];  // [2][CM_GROUPS][F] + [F]
  const int u = blockIdx.x;
  const int64_t t0 = frame_off[u], T = max_frames > 0 ? min(frame_off[u + 1] - t0, max_frames) : frame_off[u + 1] - t0;
  const int c = threadIdx.x % F, gi = threadIdx.x / F;
  double s = 0.0, q = 0.0;
  if (gi < CM_GROUPS) {
    for (int64_t t = gi; t < T; t += CM_GROUPS) {
      const double v = feat[(t0 + t) * F + c];
      s += v;
      q += v * v;
    }
    sh[gi * F + c] = s;
    sh[(CM_GROUPS + gi) * F + c] = q;
  }
  __syncthreads();
  double* norm_mean = sh + 2 * CM_GROUPS * F;
  if (threadIdx.x < F) {
    double ss = 0.0, qq = 0.0;
    for (int g2 = 0; g2 < CM_GROUPS; ++g2) {
      ss += sh[g2 * F + c];
      qq += sh[(CM_GROUPS + g2) * F + c];
    }
    const float m = T > 0 ? (float)(ss / (double)T) : 0.f;
    const float sq = T > 0 ? (float)(qq / (double)T) : 0.f;
    const float var = sq - m * m;
    const float is = norm_vars ? 1.f / sqrtf(fmaxf(var, 1e-10f)) : 1.f;
    mean[(int64_t)u * F + c] = norm_means ? m : 0.f;
    istd[(int64_t)u * F + c] = is;
    norm_mean[c] = (double)((norm_means ? 0.f : m) * is);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int k = 0; k < F; ++k) a += norm_mean[k];
    fill[u] = (float)(a / (double)F);
  }
}

// ---------------------------------------------------------------- normalise + SpecAugment + pad
// out[u, t, c] for t < min(T_u, Tmax): (x - mean)*istd, overwritten by fill[u] inside a frequency / time mask;
// pad_value for t >= T_u.  masks: int32[U, 8] = (f0,f, f0,f, t0,t, t0,t), NULL when SpecAugment is off.
template <typename TO>
__global__ void feature_finalize_kernel(const float* __restrict__ feat, const int64_t* __restrict__ frame_off,
                                        const float* __restrict__ mean, const float* __restrict__ istd,
                                        const float* __restrict__ fill, const int32_t* __restrict__ masks,
                                        TO* __restrict__ out, int64_t U, int64_t Tmax, int F, float pad_value,
                                        const int64_t* __restrict__ crop_t) {
  // crop_t (device scalar, optional): positions t >= *crop_t of EVERY utterance are 0 instead of pad_value - a batch padded to
  // a bucket length then looks to the sub-sampler's convolutions like the reference's batch, which is cropped to its longest
  // utterance and zero-padded by nn.Conv1d beyond (encoders.py:356-359)
  const int64_t total = U * Tmax * F, crop = crop_t ? *crop_t : Tmax;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = i / (Tmax * F), rem = i - u * Tmax * F, t = rem / F;
    const int c = (int)(rem - t * F);
    const int64_t t0 = frame_off[u], T = frame_off[u + 1] - t0;
    float v = t < crop ? pad_value : 0.f;
    if (t < T) {
      v = feat[(t0 + t) * F + c];
      if (mean) v = (v - mean[u * F + c]) * istd[u * F + c];
      if (masks) {
        const int32_t* m = masks + u * 8;
        const bool hit = (m[1] > 0 && c >= m[0] && c < m[0] + m[1]) || (m[3] > 0 && c >= m[2] && c < m[2] + m[3]) ||
                         (m[5] > 0 && t >= m[4] && t < m[4] + m[5]) || (m[7] > 0 && t >= m[6] && t < m[6] + m[7]);
        if (hit) v = fill[u];
      }
    }
    io<TO>::st(out + i, v);
  }
}

// ---------------------------------------------------------------- CMVN / SpecAugment on the ragged features, in place
// The general form of what feature_finalize fuses for the configured case (CMVN before SpecAugment, two masks of each kind):
// x[t, c] of utterance u becomes (x - mean[u, c]) * istd[u, c] (mean given), then fill[u] where (t, c) lies in one of the
// n_freq frequency masks or n_time time masks of the utterance.  masks: int32[U, n_freq + n_time, 2] = (start, width),
// frequency masks first.  With it the other orders are compositions: CMVN(before=False) = transform(masks) -> cmvn_stats ->
// feature_finalize(mean, istd); any number of masks = cmvn_stats -> transform(mean, istd, masks) -> feature_finalize.
__global__ void feature_transform_kernel(float* __restrict__ feat, const int64_t* __restrict__ frame_off, int F,
                                         const float* __restrict__ mean, const float* __restrict__ istd,
                                         const float* __restrict__ fill, const int32_t* __restrict__ masks, int n_freq, int n_time) {
  const int u = blockIdx.y;
  const int64_t t0 = frame_off[u], n = (frame_off[u + 1] - t0) * F;
  const int32_t* m = masks ? masks + (int64_t)u * 2 * (n_freq + n_time) : nullptr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / F;
    const int c = (int)(i - t * F);
    float v = feat[t0 * F + i];
    if (mean) v = (v - mean[(int64_t)u * F + c]) * istd[(int64_t)u * F + c];
    if (m) {
      bool hit = false;
      for (int k = 0; k < n_freq; ++k) hit = hit || (m[2 * k + 1] > 0 && c >= m[2 * k] && c < m[2 * k] + m[2 * k + 1]);
      for (int k = n_freq; k < n_freq + n_time; ++k) hit = hit || (m[2 * k + 1] > 0 && t >= m[2 * k] && t < m[2 * k] + m[2 * k + 1]);
      if (hit) v = fill[u];
    }
    feat[t0 * F + i] = v;
  }
}

}  // namespace

extern "C" int js2t_feature_transform(float* feat, const int64_t* frame_off, int32_t U, int32_t F, const float* mean, const float* istd,
                                      const float* fill, const int32_t* masks, int32_t n_freq, int32_t n_time, js2t_stream stream) {
  if (U == 0) return JS2T_OK;
  JS2T_CHECK(feat && frame_off && F > 0, "feature_transform: null pointer");
  JS2T_CHECK((mean == nullptr) == (istd == nullptr), "feature_transform: mean and istd go together");
  JS2T_CHECK(!masks || (fill && n_freq >= 0 && n_time >= 0 && n_freq + n_time > 0), "feature_transform: masks need fill values and counts");
  JS2T_CHECK(U <= 65535, "feature_transform: at most 65535 utterances per call");
  hipLaunchKernelGGL(feature_transform_kernel, dim3(16, (unsigned)U), dim3(256), 0, (hipStream_t)stream, feat, frame_off, F, mean, istd, fill,
                     masks, masks ? n_freq : 0, masks ? n_time : 0);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_fbank(const float* wave, const int64_t* sample_off, const int64_t* frame_off, int32_t U,
                          int64_t total_frames, const float* window, const float* tw_re, const float* tw_im,
                          const int32_t* mel_start, const int32_t* mel_len, const int32_t* mel_woff, const float* mel_w,
                          float* out, int32_t win_len, int32_t shift, int32_t n_fft, int32_t n_mel, float scale, float preemph,
                          float log_floor, js2t_stream stream) {
  if (total_frames == 0 || U == 0) return JS2T_OK;
  JS2T_CHECK(wave && sample_off && frame_off && window && tw_re && tw_im && mel_start && mel_len && mel_woff && mel_w && out,
             "fbank: null pointer");
  JS2T_CHECK(n_fft >= 64 && n_fft <= FB_MAX_FFT && (n_fft & (n_fft - 1)) == 0, "fbank: n_fft must be a power of two in [64,%d]",
             FB_MAX_FFT);
  JS2T_CHECK(win_len > 0 && win_len <= n_fft && shift > 0 && n_mel > 0, "fbank: bad frame geometry");
  if (n_fft == 512 && win_len <= 512 && n_mel <= 1024) {  // the Kaldi defaults at 16 kHz: FFT in registers
    int64_t grid = cdiv(total_frames, FB_WAVES);
    if (grid > 2048) grid = 2048;  // 8 blocks per CU; waves walk the frames with a grid stride
    hipLaunchKernelGGL(fbank512_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, wave, sample_off, frame_off, U, window,
                       tw_re, tw_im, mel_start, mel_len, mel_woff, mel_w, out, win_len, shift, n_mel, scale, preemph, log_floor);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  int lg = 0;
  while ((1 << lg) < n_fft) ++lg;
  hipLaunchKernelGGL(fbank_kernel, dim3(cdiv(total_frames, FB_WAVES)), dim3(256), 0, (hipStream_t)stream, wave, sample_off,
                     frame_off, U, window, tw_re, tw_im, mel_start, mel_len, mel_woff, mel_w, out, win_len, shift, n_fft, lg,
                     n_mel, scale, preemph, log_floor);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_cmvn_stats(const float* feat, const int64_t* frame_off, int32_t U, int32_t F, float* mean, float* istd,
                               float* fill, int32_t norm_means, int32_t norm_vars, int64_t max_frames, js2t_stream stream) {
  if (U == 0) return JS2T_OK;
  JS2T_CHECK(feat && frame_off && mean && istd && fill, "cmvn_stats: null pointer");
  JS2T_CHECK(F > 0 && F <= 256, "cmvn_stats: 1..256 feature bins supported");
  const int threads = ((CM_GROUPS * F + 63) / 64) * 64;
  JS2T_CHECK(threads <= 1024, "cmvn_stats: too many feature bins");
  const size_t lds = sizeof(double) * (2 * CM_GROUPS * F + F);
  hipLaunchKernelGGL(cmvn_stats_kernel, dim3(U), dim3(threads), lds, (hipStream_t)stream, feat, frame_off, F, mean, istd, fill,
                     norm_means, norm_vars, max_frames);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int64_t js2t_cmvn_stats_workspace(int32_t U, int32_t F) { return (int64_t)U * CM_SPLIT * 2 * F; }

// the same through eight blocks per utterance; workspace: f64[js2t_cmvn_stats_workspace(U, F)] from the caller
extern "C" int js2t_cmvn_stats_ws(const float* feat, const int64_t* frame_off, int32_t U, int32_t F, float* mean, float* istd, float* fill,
                                  int32_t norm_means, int32_t norm_vars, int64_t max_frames, double* workspace, js2t_stream stream) {
  if (U == 0) return JS2T_OK;
  JS2T_CHECK(feat && frame_off && mean && istd && fill, "cmvn_stats: null pointer");
  JS2T_CHECK(F > 0 && F <= 256, "cmvn_stats: 1..256 feature bins supported");
  const int threads = ((CM_GROUPS * F + 63) / 64) * 64;
  JS2T_CHECK(threads <= 1024, "cmvn_stats: too many feature bins");
  JS2T_CHECK(workspace, "cmvn_stats_ws: workspace required (js2t_cmvn_stats_workspace doubles)");
  hipLaunchKernelGGL(cmvn_partial_kernel, dim3(U, CM_SPLIT), dim3(threads), sizeof(double) * 2 * CM_GROUPS * F, (hipStream_t)stream, feat,
                     frame_off, F, workspace, max_frames);
  JS2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(cmvn_finish_kernel, dim3(U), dim3(((F + 63) / 64) * 64), sizeof(double) * F, (hipStream_t)stream, workspace,
                     frame_off, F, mean, istd, fill, norm_means, norm_vars, max_frames);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_feature_finalize(const float* feat, const int64_t* frame_off, const float* mean, const float* istd,
                                     const float* fill, const int32_t* masks, void* out, int out_dt, int64_t U, int64_t Tmax,
                                     int32_t F, float pad_value, js2t_stream stream) {
  return js2t_feature_finalize_crop(feat, frame_off, mean, istd, fill, masks, out, out_dt, U, Tmax, F, pad_value, nullptr, stream);
}

extern "C" int js2t_feature_finalize_crop(const float* feat, const int64_t* frame_off, const float* mean, const float* istd,
                                          const float* fill, const int32_t* masks, void* out, int out_dt, int64_t U, int64_t Tmax,
                                          int32_t F, float pad_value, const int64_t* crop_t, js2t_stream stream) {
  if (U * Tmax * F == 0) return JS2T_OK;
  JS2T_CHECK(feat && frame_off && out, "feature_finalize: null pointer");
  JS2T_CHECK((mean == nullptr) == (istd == nullptr), "feature_finalize: mean and istd go together");
  JS2T_CHECK(!masks || fill, "feature_finalize: masks need fill values");
  int64_t g = (U * Tmax * F + 255) / 256;
  if (g > 8192) g = 8192;
  if (out_dt == JS2T_F32)
    hipLaunchKernelGGL((feature_finalize_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, feat, frame_off,
                       mean, istd, fill, masks, (float*)out, U, Tmax, F, pad_value, crop_t);
  else if (out_dt == JS2T_BF16)
    hipLaunchKernelGGL((feature_finalize_kernel<uint16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, feat,
                       frame_off, mean, istd, fill, masks, (uint16_t*)out, U, Tmax, F, pad_value, crop_t);
  else {
    js2t_set_error("feature_finalize: bad dtype");
    return JS2T_ERR_INVALID;
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
