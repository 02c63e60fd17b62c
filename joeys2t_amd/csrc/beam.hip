// Beam-search scoring kernel (search.py:562-646): for every live batch element, fuse
//   log_softmax(logits[row]) -> forbid ids (BOS/PAD/SEP/tags, UNK, EOS below the minimum length) -> + beam log-prob
//   -> / length penalty -> top-k over the k*V flattened candidates -> (score, flat index)
// into one launch.  HBM-bound: one pass over the [k, V] logits of the element per selection round (they stay in
// L2: k*V*4 B = 100-400 KB).  One 256-thread block per batch element; wavefront shuffles for every reduction.
// Selection is by total order (score descending, flat index ascending), one block-wide arg-max per rank, so the
// result does not depend on thread scheduling: beam indices are reproducible bit for bit.
#include "common.hpp"

namespace {

constexpr int BS_THREADS = 256;
constexpr int BS_MAX_BEAM = 64;
constexpr int BS_MAX_FORBID = 16;

struct ForbidList { int32_t n; int32_t ids[BS_MAX_FORBID]; };

__device__ __forceinline__ bool better(float s, int i, float bs, int bi) { return s > bs || (s == bs && i < bi); }

__global__ __launch_bounds__(BS_THREADS) void beam_step_kernel(const float* __restrict__ logits, const float* __restrict__ beam_lp,
                                                              float* __restrict__ out_scores, int64_t* __restrict__ out_ids,
                                                              float* __restrict__ out_lse, int k, int64_t V, ForbidList fb,
                                                              float len_pen, int use_pen, int normalized) {
  __shared__ float red[BS_THREADS / 64];
  __shared__ int redi[BS_THREADS / 64];
  __shared__ float lse_s[BS_MAX_BEAM];
  __shared__ float pick_s;
  __shared__ int pick_i;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* lg = logits + (int64_t)b * k * V;
  // 1. row log-sum-exp for the k hypotheses of this element (input already log-probabilities: nothing to take off)
  if (normalized && t < k) {
    lse_s[t] = 0.f;
    out_lse[(int64_t)b * k + t] = 0.f;
  }
  for (int r = 0; r < (normalized ? 0 : k); ++r) {
    const float* row = lg + (int64_t)r * V;
    float mx = -INFINITY;
    for (int64_t v = t; v < V; v += BS_THREADS) mx = fmaxf(mx, row[v]);
    mx = block_max(mx, red);
    float s = 0.f;
    for (int64_t v = t; v < V; v += BS_THREADS) s += __expf(row[v] - mx);
    s = block_sum(s, red);
    if (t == 0) {
      lse_s[r] = mx + __logf(s);
      out_lse[(int64_t)b * k + r] = lse_s[r];
    }
  }
  __syncthreads();
  // 2. k rounds of arg-max under the (score desc, index asc) order, each bounded by the previous pick
  float prev_s = INFINITY;
  int prev_i = -1;
  const int64_t total = (int64_t)k * V;
  for (int rank = 0; rank < k; ++rank) {
    float bs = -INFINITY;
    int bi = 0x7fffffff;
    for (int64_t c = t; c < total; c += BS_THREADS) {
      const int r = (int)(c / V);
      const int v = (int)(c - (int64_t)r * V);
      float s = lg[c] - lse_s[r];
      for (int f = 0; f < fb.n; ++f)
        if (v == fb.ids[f]) s = -INFINITY;
      s += beam_lp[(int64_t)b * k + r];
      if (use_pen) s = s / len_pen;  // true division, as curr_scores /= length_penalty (search.py:628)
      // candidate must come strictly after the previous pick in the total order
      const bool after = s < prev_s || (s == prev_s && (int)c > prev_i);
      if (after && better(s, (int)c, bs, bi)) { bs = s; bi = (int)c; }
    }
    // block arg-max
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (better(os, oi, bs, bi)) { bs = os; bi = oi; }
    }
    if ((t & 63) == 0) { red[t >> 6] = bs; redi[t >> 6] = bi; }
    __syncthreads();
    if (t == 0) {
      float fs = red[0];
      int fi = redi[0];
      for (int w = 1; w < BS_THREADS / 64; ++w)
        if (better(red[w], redi[w], fs, fi)) { fs = red[w]; fi = redi[w]; }
      pick_s = fs;
      pick_i = fi;
      out_scores[(int64_t)b * k + rank] = fs;
      out_ids[(int64_t)b * k + rank] = fi == 0x7fffffff ? 0 : fi;
    }
    __syncthreads();
    prev_s = pick_s;
    prev_i = pick_i;
    __syncthreads();
  }
}


// Beams of at most BS_LK hypotheses: ONE pass over the k*V scores.  Every thread keeps the BS_LK best candidates of its
// strided share in registers (sorted insertion under the same total order); the k winners are then drawn by k block-wide
// arg-max rounds over the heads of those lists - a thread can lose at most BS_LK >= k entries, so the result is exactly
// the k best of all k*V.  Row log-sum-exps: one wave per row, single online pass, no block barrier.
constexpr int BS_LK = 8;

__global__ __launch_bounds__(BS_THREADS) void beam_step_fast_kernel(const float* __restrict__ logits, const float* __restrict__ beam_lp,
                                                                   float* __restrict__ out_scores, int64_t* __restrict__ out_ids,
                                                                   float* __restrict__ out_lse, int k, int V, ForbidList fb, float len_pen,
                                                                   int use_pen, int normalized, int n_pick) {
  __shared__ float red[BS_THREADS / 64];
  __shared__ int redi[BS_THREADS / 64];
  __shared__ int redt[BS_THREADS / 64];
  __shared__ float lse_s[BS_LK];
  __shared__ float blp_s[BS_LK];
  __shared__ int pick_t;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  const float* lg = logits + (int64_t)b * k * V;
  if (normalized && t < k) {
    lse_s[t] = 0.f;
    out_lse[(int64_t)b * k + t] = 0.f;
    blp_s[t] = beam_lp[(int64_t)b * k + t];
  }
  for (int r = w; r < (normalized ? 0 : k); r += BS_THREADS / 64) {
    const float* row = lg + (int64_t)r * V;
    float mx = -INFINITY, sm = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float x = row[v];
      if (x > mx) {
        sm = sm * __expf(mx - x) + 1.f;
        mx = x;
      } else {
        sm += __expf(x - mx);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(sm, o, 64);
      const float nm = fmaxf(mx, om);
      sm = (mx == -INFINITY ? 0.f : sm * __expf(mx - nm)) + (om == -INFINITY ? 0.f : os * __expf(om - nm));
      mx = nm;
    }
    if (lane == 0) {
      lse_s[r] = mx + __logf(sm);
      out_lse[(int64_t)b * k + r] = lse_s[r];
      blp_s[r] = beam_lp[(int64_t)b * k + r];
    }
  }
  __syncthreads();
  float ls[BS_LK];
  int li[BS_LK];
#pragma unroll
  for (int i = 0; i < BS_LK; ++i) { ls[i] = -INFINITY; li[i] = 0x7fffffff; }
  for (int r = 0; r < k; ++r) {
    const float lse = lse_s[r], blp = blp_s[r];
    const float* row = lg + (int64_t)r * V;
    for (int v = t; v < V; v += BS_THREADS) {
      float sc = row[v] - lse;
      for (int f = 0; f < fb.n; ++f)
        if (v == fb.ids[f]) sc = -INFINITY;
      sc += blp;
      if (use_pen) sc = sc / len_pen;  // true division, as curr_scores /= length_penalty (search.py:628)
      const int c = r * V + v;
      if (better(sc, c, ls[BS_LK - 1], li[BS_LK - 1])) {  // enters the list: bubble it up
        ls[BS_LK - 1] = sc;
        li[BS_LK - 1] = c;
#pragma unroll
        for (int i = BS_LK - 1; i > 0; --i) {
          if (better(ls[i], li[i], ls[i - 1], li[i - 1])) {
            const float ts = ls[i]; ls[i] = ls[i - 1]; ls[i - 1] = ts;
            const int ti = li[i]; li[i] = li[i - 1]; li[i - 1] = ti;
          }
        }
      }
    }
  }
  // n_pick (the beam search: k) rounds over the list heads
  for (int rank = 0; rank < n_pick; ++rank) {
    float bs = ls[0];
    int bi = li[0], bt = t;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64);
      const int oi = __shfl_xor(bi, o, 64), ot = __shfl_xor(bt, o, 64);
      if (better(os, oi, bs, bi)) { bs = os; bi = oi; bt = ot; }
    }
    if (lane == 0) { red[w] = bs; redi[w] = bi; redt[w] = bt; }
    __syncthreads();
    if (t == 0) {
      float fs = red[0];
      int fi = redi[0], ft = redt[0];
      for (int ww = 1; ww < BS_THREADS / 64; ++ww)
        if (better(red[ww], redi[ww], fs, fi)) { fs = red[ww]; fi = redi[ww]; ft = redt[ww]; }
      pick_t = ft;
      out_scores[(int64_t)b * n_pick + rank] = fs;
      out_ids[(int64_t)b * n_pick + rank] = fi == 0x7fffffff ? 0 : fi;
    }
    __syncthreads();
    if (t == pick_t) {  // pop the winner's list
#pragma unroll
      for (int i = 0; i < BS_LK - 1; ++i) { ls[i] = ls[i + 1]; li[i] = li[i + 1]; }
      ls[BS_LK - 1] = -INFINITY;
      li[BS_LK - 1] = 0x7fffffff;
    }
    __syncthreads();
  }
}

// Repetition penalty (search.py:972-1001, after Huggingface's RepetitionPenaltyLogitsProcessor): for every token id in
// tokens[row, :] the row's log-probability x becomes x * penalty if x < 0 else x / penalty.  gather -> scale -> scatter like
// the reference: every occurrence reads the ORIGINAL value (staged in LDS before anything is written), so a token that
// occurs several times is penalised once.  One block per row.
__global__ __launch_bounds__(256) void rep_penalty_kernel(float* __restrict__ logp, const int64_t* __restrict__ tokens, int L, int64_t V,
                                                         float penalty) {
  extern __shared__ float orig[];
  float* row = logp + (int64_t)blockIdx.x * V;
  const int64_t* tk = tokens + (int64_t)blockIdx.x * L;
  for (int i = threadIdx.x; i < L; i += 256) orig[i] = row[tk[i]];
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += 256) {
    const float x = orig[i];
    row[tk[i]] = x < 0.f ? x * penalty : x / penalty;
  }
}

// logp[rows[i], cols[i]] = value: banned n-grams (search.py:966-969), forbidden ids ahead of a forced token (:590-601) and
// the forced tokens of a decoder prompt themselves (:614-618)
__global__ void logp_set_kernel(float* __restrict__ logp, const int64_t* __restrict__ rows, const int64_t* __restrict__ cols, int64_t n,
                                int64_t V, float value) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) logp[rows[i] * V + cols[i]] = value;
}

int beam_step_launch(const float* logits, const float* beam_log_probs, float* out_scores, int64_t* out_ids, float* out_lse,
                     int64_t n_batch, int32_t beam, int64_t V, const int32_t* forbid_ids, int32_t n_forbid, float length_penalty,
                     int normalized, js2t_stream stream);

// ---------------------------------------------------------------------------------------------------------------------------------
// Joint CTC / attention decoding (EXTENSION: the reference returns ctc_out for return_type="decode_ctc", model.py:162-166,
// and has no consumer).  One step of the CTC prefix score of Watanabe et al., "Hybrid CTC/Attention Architecture for End-to-End
// Speech Recognition" (IEEE JSTSP 2017, Algorithm 2), for every (live hypothesis, candidate token) pair - one thread each, the
// recursion over the utterance's frames is sequential:
//   r_n[t] = logaddexp(r_n[t-1], phi[t-1]) + x[t, c]      phi = r_n + r_b of the hypothesis, or its r_b alone when c repeats its last label
//   r_b[t] = logaddexp(r_n[t-1], r_b[t-1]) + x[t, blank]
//   psi    = logaddexp over t of (phi[t-1] + x[t, c])      (c = EOS: r_n[T-1] + r_b[T-1] of the hypothesis - it ends here)
// and the step's local score  (1 - w) * log p_att(c) + w * (psi - psi of the hypothesis)  that the beam selection adds to the
// accumulated one.  log 0 is carried as -1e30 (no inf - inf).  oracle: oracle/s2t_oracle.py ctc_prefix_score, pinned by enumeration.
constexpr float CTC_LOG0 = -1.0e30f;
__device__ __forceinline__ float ctc_lae(float a, float b) {
  const float m = fmaxf(a, b);
  return m <= 0.5f * CTC_LOG0 ? CTC_LOG0 : m + log1pf(__expf(-fabsf(a - b)));
}
__global__ __launch_bounds__(64) void ctc_prefix_step_kernel(const float* __restrict__ x, const int64_t* __restrict__ in_len,
                                                            const float* __restrict__ r_prev, const int64_t* __restrict__ last_tok,
                                                            const int64_t* __restrict__ cand, const float* __restrict__ cand_lp,
                                                            const float* __restrict__ psi_prev, float* __restrict__ local,
                                                            float* __restrict__ psi_out, float* __restrict__ r_new, int64_t n_pairs, int k, int C,
                                                            int T, int64_t V, int n_out, int blank, int eos, float w) {
  const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (idx >= n_pairs) return;
  const int64_t row = idx / C, b = row / k;
  const int Tb = (int)min((int64_t)T, in_len[b]);
  const float* xb = x + b * (int64_t)T * V;
  const float* rp = r_prev + row * (int64_t)T * 2;
  float* rn = r_new + idx * (int64_t)T * 2;
  const int64_t c = cand[idx];
  const bool repeat = n_out > 0 && c == last_tok[row];
  for (int t = 0; t < T; ++t) rn[2 * t] = CTC_LOG0, rn[2 * t + 1] = CTC_LOG0;
  float r_n = CTC_LOG0, r_b = CTC_LOG0, psi = CTC_LOG0;
  const int start = max(n_out, 1);
  if (n_out == 0 && Tb > 0) {
    r_n = xb[c];
    rn[0] = r_n;
  }
  if (start - 1 < Tb) psi = n_out == 0 ? r_n : CTC_LOG0;  // r[start - 1, 0]: only the first label can have been emitted by frame 0
  const int t0 = min(start - 1, T - 1);  // (a hypothesis longer than the input has frames: the loop below does not run)
  float pn = rp[2 * t0], pb = rp[2 * t0 + 1];  // the hypothesis' variables at t - 1
  for (int t = start; t < Tb; ++t) {
    const float phi = repeat ? pb : ctc_lae(pn, pb);
    const float xt = xb[(int64_t)t * V + c], xblank = xb[(int64_t)t * V + blank];
    const float nn = ctc_lae(r_n, phi) + xt;
    const float nb = ctc_lae(r_n, r_b) + xblank;
    psi = ctc_lae(psi, phi + xt);
    r_n = fmaxf(nn, CTC_LOG0), r_b = fmaxf(nb, CTC_LOG0);
    rn[2 * t] = r_n, rn[2 * t + 1] = r_b;
    pn = rp[2 * t], pb = rp[2 * t + 1];
  }
  if (c == eos) psi = Tb > 0 ? ctc_lae(rp[2 * (Tb - 1)], rp[2 * (Tb - 1) + 1]) : CTC_LOG0;
  if (c == blank) psi = CTC_LOG0;
  psi = fmaxf(psi, CTC_LOG0);
  psi_out[idx] = psi;
  const float att = cand_lp[idx], pp = psi_prev[row];
  float loc = (1.f - w) * att;
  if (w > 0.f) loc = (psi <= 0.5f * CTC_LOG0 || pp <= 0.5f * CTC_LOG0) ? -INFINITY : loc + w * (psi - pp);
  if (!(att > -INFINITY)) loc = -INFINITY;
  local[idx] = loc;
}

// Round 6: the same step with the operands staged through LDS.  The recursion is a chain over the frames, but nothing it READS
// depends on the chain - the thread-per-pair kernel above still waited out a cold 20 KB-strided gather (and two loads of the
// hypothesis' variables) in every one of its T iterations: 375 dependent ~1.1 us round trips = 0.44 ms on every 1 ms decode step
// (bench.py decode_beam5.joint_ctc, round 5).  Here a block owns ONE hypothesis: its 64 threads first fetch x[t, c] of the C
// candidates, x[t, blank] and the hypothesis' (r_n, r_b) for all frames - (C + 3) T independent loads in flight at once - into
// LDS, then C threads walk the chain from LDS (no memory latency left on it: ~100 cycles of logaddexp arithmetic per frame) and
// write the new variables as they go.  That alone changed nothing (557 against 492 us, tools/ctc_prefix_bench.py): the compiler had
// the round-5 kernel's loads ahead of their use already - what a frame really costs is the CHAIN's arithmetic, three logaddexp of
// log1pf(expf()) each (the accurate library forms: ~300 dependent vector instructions per frame on a wave with eight live lanes).
// Here logaddexp is max + log(1 + exp(-|a - b|)) on the hardware's v_exp_f32 / v_log_f32 (ctc_lae_fast: ~10 instructions; absolute
// error ~1e-7 per call, against sums of magnitude 10^3 whose f32 spacing is 1e-4): same results within f32 rounding of the sums
// (tests/test_hip_search_ctc.py: the oracle's recursion at 1e-4, the round-5 kernel at full size, ids of the golden models bit-exact).
__device__ __forceinline__ float ctc_lae_fast(float a, float b) {
  const float m = fmaxf(a, b);
  return m <= 0.5f * CTC_LOG0 ? CTC_LOG0 : m + __logf(1.f + __expf(-fabsf(a - b)));
}
__global__ __launch_bounds__(64) void ctc_prefix_step_lds_kernel(const float* __restrict__ x, const int64_t* __restrict__ in_len,
                                                                const float* __restrict__ r_prev, const int64_t* __restrict__ last_tok,
                                                                const int64_t* __restrict__ cand, const float* __restrict__ cand_lp,
                                                                const float* __restrict__ psi_prev, float* __restrict__ local,
                                                                float* __restrict__ psi_out, float* __restrict__ r_new, int k, int C, int T,
                                                                int64_t V, int n_out, int blank, int eos, float w) {
  // LDS: [C + 1][T] log-probabilities x[t, c_j] (row C: the blank), [T] phi of the hypothesis (logaddexp of its two variables),
  // [T][2] the variables themselves, [C] partial prefix scores
  extern __shared__ float ctc_lds[];
  const int64_t row = blockIdx.x, b = row / k;
  const int tid = threadIdx.x;
  const int Tb = (int)min((int64_t)T, in_len[b]);
  const float* xb = x + b * (int64_t)T * V;
  const float2* rp = (const float2*)(r_prev + row * (int64_t)T * 2);
  float* xs = ctc_lds;
  float* phis = ctc_lds + (int64_t)(C + 1) * T;
  float2* rs = (float2*)(phis + T);
  const int start = max(n_out, 1);
  for (int j = 0; j <= C; ++j) {
    const int64_t c = j < C ? cand[row * C + j] : (int64_t)blank;
    for (int t = start + tid; t < Tb; t += 64) xs[j * T + t] = xb[(int64_t)t * V + c];
  }
  for (int t = tid; t < T; t += 64) {
    const float2 v = rp[t];
    rs[t] = v;
    phis[t] = ctc_lae_fast(v.x, v.y);
  }
  // the variables no frame of the chain writes: log 0 (frame 0 of a first label is written by its chain thread below)
  for (int j = 0; j < C; ++j) {
    float2* rn = (float2*)(r_new + (row * C + j) * (int64_t)T * 2);
    for (int t = tid; t < T; t += 64)
      if ((t < start || t >= Tb) && !(t == 0 && n_out == 0 && Tb > 0)) rn[t] = make_float2(CTC_LOG0, CTC_LOG0);
  }
  __syncthreads();
  // The prefix score psi = logaddexp over t of (phi[t - 1] + x[t, c]) reads nothing of the chain: 64 / C lanes per candidate sum
  // their frames and leave the partial sums in LDS for the candidate's chain thread - the chain keeps two independent logaddexp per frame.
  const int lanes_per = 64 / C, j_mine = tid / lanes_per, sub = tid % lanes_per;  // C <= 64 (host check); lanes beyond C * lanes_per idle
  float* parts = (float*)(rs + T);  // [C][lanes_per] <= 64 values
  if (j_mine < C) {
    const bool rep_mine = n_out > 0 && cand[row * C + j_mine] == last_tok[row];
    float part = CTC_LOG0;
    for (int t = start + sub; t < Tb; t += lanes_per) part = ctc_lae_fast(part, (rep_mine ? rs[t - 1].y : phis[t - 1]) + xs[j_mine * T + t]);
    parts[j_mine * lanes_per + sub] = part;
  }
  __syncthreads();
  if (tid >= C) return;
  const int64_t idx = row * C + tid;
  const int64_t c = cand[idx];
  const bool repeat = n_out > 0 && c == last_tok[row];
  float2* rn = (float2*)(r_new + idx * (int64_t)T * 2);
  const float* xc = xs + tid * T;
  const float* xblk = xs + C * T;
  float r_n = CTC_LOG0, r_b = CTC_LOG0, psi = CTC_LOG0;
  for (int q = 0; q < lanes_per; ++q) psi = ctc_lae_fast(psi, parts[tid * lanes_per + q]);
  if (n_out == 0 && Tb > 0) {
    r_n = xb[c];
    rn[0] = make_float2(r_n, CTC_LOG0);
  }
  if (start - 1 < Tb && n_out == 0) psi = ctc_lae_fast(psi, r_n);  // r[start - 1, 0]: only the first label can have been emitted by frame 0
  // the chain: operands of frame t + 1 are read from LDS while frame t is computed
  float phi = 0.f, xt = 0.f, xblank = 0.f;
  if (start < Tb) phi = repeat ? rs[start - 1].y : phis[start - 1], xt = xc[start], xblank = xblk[start];
  for (int t = start; t < Tb; ++t) {
    const int tn = min(t + 1, Tb - 1);
    const float phi_n = repeat ? rs[tn - 1].y : phis[tn - 1], xt_n = xc[tn], xblank_n = xblk[tn];
    const float nn = ctc_lae_fast(r_n, phi) + xt;
    const float nb = ctc_lae_fast(r_n, r_b) + xblank;
    r_n = fmaxf(nn, CTC_LOG0), r_b = fmaxf(nb, CTC_LOG0);
    rn[t] = make_float2(r_n, r_b);
    phi = phi_n, xt = xt_n, xblank = xblank_n;
  }
  if (c == eos) psi = Tb > 0 ? ctc_lae_fast(rs[Tb - 1].x, rs[Tb - 1].y) : CTC_LOG0;
  if (c == blank) psi = CTC_LOG0;
  psi = fmaxf(psi, CTC_LOG0);
  psi_out[idx] = psi;
  const float att = cand_lp[idx], pp = psi_prev[row];
  float loc = (1.f - w) * att;
  if (w > 0.f) loc = (psi <= 0.5f * CTC_LOG0 || pp <= 0.5f * CTC_LOG0) ? -INFINITY : loc + w * (psi - pp);
  if (!(att > -INFINITY)) loc = -INFINITY;
  local[idx] = loc;
}

}  // namespace

// The n_pick best tokens of EVERY row (log-softmax, forbidden ids masked, + the row's entry of row_scores): the candidate
// pre-selection of joint CTC / attention decoding - the fast beam kernel with one row per entry
extern "C" int js2t_beam_pick(const float* logits, const float* row_scores, float* out_scores, int64_t* out_ids, float* out_lse, int64_t rows,
                              int32_t n_pick, int64_t V, const int32_t* forbid_ids, int32_t n_forbid, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(logits && row_scores && out_scores && out_ids && out_lse, "beam_pick: null pointer");
  JS2T_CHECK(n_pick >= 1 && n_pick <= BS_LK && n_pick <= V, "beam_pick: 1..%d candidates (at most the vocabulary)", BS_LK);
  JS2T_CHECK(n_forbid >= 0 && n_forbid <= BS_MAX_FORBID && (n_forbid == 0 || forbid_ids), "beam_pick: at most %d forbidden ids", BS_MAX_FORBID);
  JS2T_CHECK(V < 0x7fffffff, "beam_pick: vocabulary too large");
  ForbidList fb;
  fb.n = n_forbid;
  for (int i = 0; i < BS_MAX_FORBID; ++i) fb.ids[i] = i < n_forbid ? forbid_ids[i] : -1;
  hipLaunchKernelGGL(beam_step_fast_kernel, dim3((unsigned)rows), dim3(BS_THREADS), 0, (hipStream_t)stream, logits, row_scores, out_scores,
                     out_ids, out_lse, 1, (int)V, fb, 1.f, 0, 0, n_pick);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

static bool g_ctc_prefix_thread_per_pair = false;  // test hook: the round-5 kernel (also taken when the frames do not fit 64 KB of LDS)
extern "C" void js2t_debug_ctc_prefix_thread_per_pair(int on) { g_ctc_prefix_thread_per_pair = on != 0; }

extern "C" int js2t_ctc_prefix_step(const float* ctc_log_probs, const int64_t* in_len, const float* r_prev, const int64_t* last_tok,
                                    const int64_t* cand, const float* cand_lp, const float* psi_prev, float* local, float* psi_out,
                                    float* r_new, int64_t rows, int32_t beam, int32_t n_cand, int32_t T, int64_t V, int32_t n_out,
                                    int32_t blank, int32_t eos, float weight, js2t_stream stream) {
  if (rows == 0) return JS2T_OK;
  JS2T_CHECK(ctc_log_probs && in_len && r_prev && last_tok && cand && cand_lp && psi_prev && local && psi_out && r_new, "ctc_prefix_step: null pointer");
  JS2T_CHECK(beam >= 1 && n_cand >= 1 && T >= 1 && V >= 1 && rows % beam == 0 && n_out >= 0, "ctc_prefix_step: bad sizes");
  JS2T_CHECK(blank >= 0 && blank < V && eos >= 0 && eos < V && weight >= 0.f && weight <= 1.f, "ctc_prefix_step: bad blank / eos / weight");
  const int64_t n = rows * n_cand;
  const size_t lds = ((size_t)(n_cand + 1) * T + 3 * (size_t)T + 64) * sizeof(float);
  if (n_cand <= 64 && lds <= 64 * 1024 && !g_ctc_prefix_thread_per_pair) {  // a block per hypothesis, operands through LDS (the default)
    hipLaunchKernelGGL(ctc_prefix_step_lds_kernel, dim3((unsigned)rows), dim3(64), lds, (hipStream_t)stream, ctc_log_probs, in_len, r_prev, last_tok,
                       cand, cand_lp, psi_prev, local, psi_out, r_new, beam, n_cand, T, V, n_out, blank, eos, weight);
    JS2T_LAUNCH_CHECK();
    return JS2T_OK;
  }
  hipLaunchKernelGGL(ctc_prefix_step_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, ctc_log_probs, in_len, r_prev,
                     last_tok, cand, cand_lp, psi_prev, local, psi_out, r_new, n, beam, n_cand, T, V, n_out, blank, eos, weight);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_rep_penalty(float* log_probs, const int64_t* tokens, int64_t rows, int64_t V, int64_t L, float penalty,
                                js2t_stream stream) {
  if (rows == 0 || L == 0) return JS2T_OK;
  JS2T_CHECK(log_probs && tokens && V > 0, "rep_penalty: null pointer");
  JS2T_CHECK(L <= 8192, "rep_penalty: at most 8192 tokens per row");
  JS2T_CHECK(penalty > 0.f, "rep_penalty: penalty must be positive");
  hipLaunchKernelGGL(rep_penalty_kernel, dim3((unsigned)rows), dim3(256), (size_t)L * sizeof(float), (hipStream_t)stream, log_probs,
                     tokens, (int)L, V, penalty);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_logp_set(float* log_probs, const int64_t* rows, const int64_t* cols, int64_t n, int64_t V, float value,
                             js2t_stream stream) {
  if (n == 0) return JS2T_OK;
  JS2T_CHECK(log_probs && rows && cols && V > 0, "logp_set: null pointer");
  hipLaunchKernelGGL(logp_set_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, log_probs, rows, cols, n, V,
                     value);
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}

extern "C" int js2t_beam_step(const float* logits, const float* beam_log_probs, float* out_scores, int64_t* out_ids,
                              float* out_lse, int64_t n_batch, int32_t beam, int64_t V, const int32_t* forbid_ids,
                              int32_t n_forbid, float length_penalty, js2t_stream stream) {
  return beam_step_launch(logits, beam_log_probs, out_scores, out_ids, out_lse, n_batch, beam, V, forbid_ids, n_forbid, length_penalty, 0,
                          stream);
}

extern "C" int js2t_beam_step_logp(const float* log_probs, const float* beam_log_probs, float* out_scores, int64_t* out_ids,
                                   float* out_lse, int64_t n_batch, int32_t beam, int64_t V, const int32_t* forbid_ids,
                                   int32_t n_forbid, float length_penalty, js2t_stream stream) {
  return beam_step_launch(log_probs, beam_log_probs, out_scores, out_ids, out_lse, n_batch, beam, V, forbid_ids, n_forbid, length_penalty,
                          1, stream);
}

namespace {
int beam_step_launch(const float* logits, const float* beam_log_probs, float* out_scores, int64_t* out_ids, float* out_lse,
                     int64_t n_batch, int32_t beam, int64_t V, const int32_t* forbid_ids, int32_t n_forbid, float length_penalty,
                     int normalized, js2t_stream stream) {
  if (n_batch == 0) return JS2T_OK;
  JS2T_CHECK(logits && beam_log_probs && out_scores && out_ids && out_lse, "beam_step: null pointer");
  JS2T_CHECK(beam >= 1 && beam <= BS_MAX_BEAM, "beam_step: beam size 1..%d", BS_MAX_BEAM);
  JS2T_CHECK(n_forbid >= 0 && n_forbid <= BS_MAX_FORBID && (n_forbid == 0 || forbid_ids), "beam_step: at most %d forbidden ids",
             BS_MAX_FORBID);
  JS2T_CHECK((int64_t)beam * V < 0x7fffffff, "beam_step: beam * vocab too large");
  ForbidList fb;
  fb.n = n_forbid;
  for (int i = 0; i < BS_MAX_FORBID; ++i) fb.ids[i] = i < n_forbid ? forbid_ids[i] : -1;
  const int use_pen = length_penalty > 0.f ? 1 : 0;
  if (beam <= BS_LK) {
    hipLaunchKernelGGL(beam_step_fast_kernel, dim3((unsigned)n_batch), dim3(BS_THREADS), 0, (hipStream_t)stream, logits, beam_log_probs,
                       out_scores, out_ids, out_lse, beam, (int)V, fb, use_pen ? length_penalty : 1.f, use_pen, normalized, beam);
  } else {
    hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)n_batch), dim3(BS_THREADS), 0, (hipStream_t)stream, logits, beam_log_probs,
                       out_scores, out_ids, out_lse, beam, V, fb, use_pen ? length_penalty : 1.f, use_pen, normalized);
  }
  JS2T_LAUNCH_CHECK();
  return JS2T_OK;
}
}  // namespace
