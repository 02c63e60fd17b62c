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

// ---------------------------------------------------------------- CMVN statistics (one block per utterance)
// mean[u,c] = mean_t x ; istd[u,c] = 1/sqrt(max(sum x^2 / T - mean^2, 1e-10)) ; fill[u] = mean of the normalised
// spectrogram (SpecAugment's mask value).  Accumulated in f64.
constexpr int CM_GROUPS = 12;  // frame groups per block: 12 x 80 bins = 960 threads share an utterance
__global__ void cmvn_stats_kernel(const float* __restrict__ feat, const int64_t* __restrict__ frame_off, int F,
                                  float* __restrict__ mean, float* __restrict__ istd, float* __restrict__ fill, int norm_means,
                                  int norm_vars, int64_t max_frames) {
  extern __shared__ double sh[];  // [2][CM_GROUPS][F] + [F]
  const int u = blockIdx.x;
  // an over-long evaluation utterance is cut to max_length BEFORE CMVN (tokenizers.py:474-487): statistics over the kept frames
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
                                        TO* __restrict__ out, int64_t U, int64_t Tmax, int F, float pad_value) {
  const int64_t total = U * Tmax * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = i / (Tmax * F), rem = i - u * Tmax * F, t = rem / F;
    const int c = (int)(rem - t * F);
    const int64_t t0 = frame_off[u], T = frame_off[u + 1] - t0;
    float v = pad_value;
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

}  // namespace

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

extern "C" int js2t_feature_finalize(const float* feat, const int64_t* frame_off, const float* mean, const float* istd,
                                     const float* fill, const int32_t* masks, void* out, int out_dt, int64_t U, int64_t Tmax,
                                     int32_t F, float pad_value, js2t_stream stream) {
  if (U * Tmax * F == 0) return JS2T_OK;
  JS2T_CHECK(feat && frame_off && out, "feature_finalize: null pointer");
  JS2T_CHECK((mean == nullptr) == (istd == nullptr), "feature_finalize: mean and istd go together");
  JS2T_CHECK(!masks || fill, "feature_finalize: masks need fill values");
  int64_t g = (U * Tmax * F + 255) / 256;
  if (g > 8192) g = 8192;
  if (out_dt == JS2T_F32)
    hipLaunchKernelGGL((feature_finalize_kernel<float>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, feat, frame_off,
                       mean, istd, fill, masks, (float*)out, U, Tmax, F, pad_value);
  else if (out_dt == JS2T_BF16)
    hipLaunchKernelGGL((feature_finalize_kernel<uint16_t>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, feat,
                       frame_off, mean, istd, fill, masks, (uint16_t*)out, U, Tmax, F, pad_value);
  else {
    js2t_set_error("feature_finalize: bad dtype");
    return JS2T_ERR_INVALID;
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
