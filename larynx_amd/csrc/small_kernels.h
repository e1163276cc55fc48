// The non-GEMM kernels of the Larynx hot path: gathers, per-column reductions
// (LayerNorm, softmax), the duration -> frame expansion, the invertible 4x4
// mixing + ActNorm, the mel transform and the int16 conversion.  All are
// HBM/L2-bound byte shufflers: one pass, coalesced along time, wave64
// reductions through __shfl_xor.
//
// Tensors are [B][C][ld] with time fastest; `len[b]` is the valid length of row
// b and everything beyond it is never read (readers mask) — that reproduces the
// reference's `* x_mask` (glow_tts/models.py:118-140) and gives the vocoder the
// per-utterance zero padding the un-batched reference has.
#pragma once
#include <hip/hip_runtime.h>
#include "prio.h"
#include <stdint.h>

namespace mi355tts {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// G1: x[b][c][t] = emb[ids[b][t]][c] * sqrt(H)           (glow_tts/models.py:119-120)
__global__ void embed_kernel(const long long* ids, int ids_ld, const int* len, const float* emb, int V, int H,
                             float scale, float* x, long long x_bs, int x_ld) {
  GLOW_PRIO();
  const int b = blockIdx.z;
  const int t = blockIdx.x * 64 + (threadIdx.x & 63);
  const int c0 = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (t >= len[b]) return;
  long long id = ids[(long long)b * ids_ld + t];
  if (id < 0) id = 0;
  if (id >= V) id = V - 1;
  for (int c = c0; c < H; c += 4 * gridDim.y) x[(long long)b * x_bs + (long long)c * x_ld + t] = emb[id * H + c] * scale;
}

// LayerNorm over channels of (x [+ res]) per time column (glow_tts/layers.py:19-28),
// eps inside the sqrt, biased variance.  pre_relu: DurationPredictor order
// conv -> ReLU -> LN (models.py:39-49); post_relu: prenet order conv -> LN -> ReLU
// (layers.py:73-80).  Block = 64 columns x 4 channel groups.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* res, const float* gamma,
                                                        const float* beta, float* y, int C, long long bs, int ld,
                                                        const int* len, int pre_relu, int post_relu, float eps) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int tl = threadIdx.x & 63;
  const int g = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  const bool valid = t < len[b];
  const float* xb = x + (long long)b * bs + t;
  const float* rb = res ? res + (long long)b * bs + t : nullptr;
  float s = 0.f;
  if (valid)
    for (int c = g; c < C; c += 4) {
      float v = xb[(long long)c * ld];
      if (rb) v += rb[(long long)c * ld];
      if (pre_relu) v = fmaxf(v, 0.f);
      s += v;
    }
  red[g][tl] = s;
  __syncthreads();
  const float mean = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
  __syncthreads();
  float q = 0.f;
  if (valid)
    for (int c = g; c < C; c += 4) {
      float v = xb[(long long)c * ld];
      if (rb) v += rb[(long long)c * ld];
      if (pre_relu) v = fmaxf(v, 0.f);
      const float d = v - mean;
      q += d * d;
    }
  red[g][tl] = q;
  __syncthreads();
  const float var = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / (float)C;
  const float rstd = rsqrtf(var + eps);
  if (!valid) return;
  float* yb = y + (long long)b * bs + t;
  for (int c = g; c < C; c += 4) {
    float v = xb[(long long)c * ld];
    if (rb) v += rb[(long long)c * ld];
    if (pre_relu) v = fmaxf(v, 0.f);
    float o = (v - mean) * rstd * gamma[c] + beta[c];
    if (post_relu) o = fmaxf(o, 0.f);
    yb[(long long)c * ld] = o;
  }
}

// Fast path for C <= 256: 16 columns x 16 channel groups per workgroup, the
// column's values live in registers (one global read, one write), group sums
// meet in LDS.
// `proj_w` != nullptr: the normalised column is not stored but contracted with proj_w[C] (+ proj_b[0]) into
// proj_y[b][t] — the duration predictor's norm_2 -> proj (1 x 1, C -> 1; glow_tts/models.py:39-49) in one launch.
__global__ __launch_bounds__(256) void layernorm16_kernel(const float* x, const float* res, const float* gamma,
                                                          const float* beta, float* y, int C, long long bs, int ld,
                                                          const int* len, int pre_relu, int post_relu, float eps,
                                                          const float* proj_w = nullptr, const float* proj_b = nullptr,
                                                          float* proj_y = nullptr, long long proj_bs = 0) {
  GLOW_PRIO();
  __shared__ float red[16][17];
  const int b = blockIdx.y;
  const int tl = threadIdx.x & 15;
  const int g = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + tl;
  const int L = len[b];
  const bool valid = t < L;
  const int tc = valid ? t : (L > 0 ? L - 1 : 0);
  const float* xb = x + (long long)b * bs + tc;
  const float* rb = res ? res + (long long)b * bs + tc : nullptr;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = g + 16 * i;
    const int cc = c < C ? c : C - 1;
    float u = xb[(long long)cc * ld];
    if (rb) u += rb[(long long)cc * ld];
    if (pre_relu) u = fmaxf(u, 0.f);
    v[i] = c < C ? u : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  red[g][tl] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) mean += red[k][tl];
  mean /= (float)C;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float d = (g + 16 * i < C) ? v[i] - mean : 0.f;
    q += d * d;
  }
  red[g][tl] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) var += red[k][tl];
  var /= (float)C;
  const float rstd = rsqrtf(var + eps);
  if (proj_w) {  // kernel-uniform
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = g + 16 * i;
      if (c < C) {
        float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
        if (post_relu) o = fmaxf(o, 0.f);
        d += o * proj_w[c];
      }
    }
    __syncthreads();
    red[g][tl] = d;
    __syncthreads();
    if (g == 0 && valid) {
      float o = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) o += red[k][tl];
      proj_y[(long long)b * proj_bs + t] = o + proj_b[0];
    }
    return;
  }
  if (!valid) return;
  float* yb = y + (long long)b * bs + t;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = g + 16 * i;
    if (c < C) {
      float o = (v[i] - mean) * rstd * gamma[c] + beta[c];
      if (post_relu) o = fmaxf(o, 0.f);
      yb[(long long)c * ld] = o;
    }
  }
}

// G4: windowed relative-position self-attention (glow_tts/attentions.py:214-264).
// qkv is [B][3H][ld] (q rows 0..H, k rows H..2H, v rows 2H..3H), head h uses
// channels h*dk..(h+1)*dk.  One wave per query row i, 4 rows per workgroup:
//   s[j] = (q_i.k_j + [|j-i|<=w] q_i.Ek[j-i+w]) / sqrt(dk);  p = softmax_j(s)
//   o[c] = sum_j p[j] v[c][j] + sum_{|r-w|<=w} p[i+r-w] Ev[r][c]
// Keys j >= len[b] carry exactly zero weight (the reference fills -1e4, whose
// softmax weight underflows to 0 in fp32).  The band is evaluated directly; the
// reference's pad/reshape skewing (attentions.py:284-335) is never materialised.
constexpr int ATT_ROWS = 4;
constexpr int ATT_MAXW = 16;   // max 2*window+1
constexpr int ATT_JCH = 64;    // keys per staged V chunk
constexpr int ATT_MAXDK = 128;

__global__ __launch_bounds__(256) void attention_kernel(const float* qkv, long long bs, int ld, const int* len, int H,
                                                        int nheads, int window, const float* ek, const float* ev,
                                                        float* out, long long out_bs, int out_ld, float* scores_ws,
                                                        int ws_ld) {
  __shared__ float rel[ATT_ROWS][ATT_MAXW];
  __shared__ float vs[ATT_MAXDK][ATT_JCH + 1];
  __shared__ float ps[ATT_ROWS][ATT_JCH];
  const int b = blockIdx.z;
  const int h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int P = len[b];
  const int i = blockIdx.x * ATT_ROWS + wave;
  const int dk = H / nheads;
  const int nrel = 2 * window + 1;
  const float scale = rsqrtf((float)dk);
  const bool row_ok = i < P;
  const float* q = qkv + (long long)b * bs + (long long)(h * dk) * ld;
  const float* k = q + (long long)H * ld;
  const float* v = k + (long long)H * ld;
  // per-(b,h,row) scratch for the score row (P can exceed what fits in LDS)
  float* srow = scores_ws + ((long long)(b * nheads + h) * gridDim.x * ATT_ROWS + (blockIdx.x * ATT_ROWS + wave)) * ws_ld;

  // relative-key logits for the 2w+1 band
  for (int r = 0; r < nrel; ++r) {
    float part = 0.f;
    if (row_ok)
      for (int c = lane; c < dk; c += 64) part += q[(long long)c * ld + i] * ek[r * dk + c];
    part = wave_sum(part);
    if (lane == 0) rel[wave][r] = part;
  }
  __syncthreads();

  // scores and running max
  float mx = -3.0e38f;
  if (row_ok) {
    for (int j = lane; j < P; j += 64) {
      float s = 0.f;
      for (int c = 0; c < dk; ++c) s += q[(long long)c * ld + i] * k[(long long)c * ld + j];
      const int r = j - i + window;
      float sl = 0.f;
      if (r >= 0 && r < nrel) sl = rel[wave][r];
      s = s * scale + sl * scale;
      srow[j] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = wave_max(mx);
  float den = 0.f;
  if (row_ok) {
    for (int j = lane; j < P; j += 64) {
      const float e = expf(srow[j] - mx);
      srow[j] = e;
      den += e;
    }
  }
  den = wave_sum(den);
  const float inv = row_ok ? 1.0f / den : 0.f;

  // o[c] accumulation: lanes over channels, V staged chunk-wise through LDS
  float acc0 = 0.f, acc1 = 0.f;  // channels lane and lane+64
  for (int j0 = 0; j0 < P; j0 += ATT_JCH) {
    __syncthreads();
    for (int e = threadIdx.x; e < dk * ATT_JCH; e += 256) {
      const int c = e / ATT_JCH, jj = e - c * ATT_JCH;
      vs[c][jj] = (j0 + jj < P) ? v[(long long)c * ld + j0 + jj] : 0.f;
    }
    if (lane < ATT_JCH) ps[wave][lane] = (row_ok && j0 + lane < P) ? srow[j0 + lane] * inv : 0.f;
    __syncthreads();
    const int jn = (P - j0 < ATT_JCH) ? P - j0 : ATT_JCH;
    for (int jj = 0; jj < jn; ++jj) {
      const float p = ps[wave][jj];
      if (lane < dk) acc0 += p * vs[lane][jj];
      if (lane + 64 < dk) acc1 += p * vs[lane + 64][jj];
    }
  }
  if (!row_ok) return;
  // relative values on the band
  for (int r = 0; r < nrel; ++r) {
    const int j = i + r - window;
    if (j < 0 || j >= P) continue;
    const float p = srow[j] * inv;
    if (lane < dk) acc0 += p * ev[r * dk + lane];
    if (lane + 64 < dk) acc1 += p * ev[r * dk + lane + 64];
  }
  float* ob = out + (long long)b * out_bs + (long long)(h * dk) * out_ld + i;
  if (lane < dk) ob[(long long)lane * out_ld] = acc0;
  if (lane + 64 < dk) ob[(long long)(lane + 64) * out_ld] = acc1;
}

// G4 on the matrix cores (the path used for P <= ATTM_MAXP): one workgroup per
// (32 query rows, head).  S = Q K^T and O^T = V P^T are v_mfma_f32_32x32x2_f32
// GEMMs whose operands come straight out of the [channel][time] qkv layout
// (lanes run along time = coalesced); the score block lives in LDS between the
// two, where the 9-wide relative-position band is added and the rows are
// soft-maxed with wave64 reductions.
constexpr int ATTM_MAXP = 768;
constexpr int ATTM_PS = ATTM_MAXP + 1;  // odd row stride: conflict-free column reads
typedef float att_floatx16 __attribute__((ext_vector_type(16)));

#ifndef ATT_STAMP
#define ATT_STAMP(n)
#endif
// The launch is a latency chain on a handful of workgroups (8 at P = 120), and at one or two waves per SIMD its
// time is the instruction count of the slowest wave (phase stamps: profiles/NOTES.md).  So: eight waves share
// every phase, nothing is masked that is never read, addresses advance by one add per load, and every global
// load whose address is known at entry is issued at entry.  EXACT: dk == 2*NK (no channel clamps at all).
// PMAXT: the longest sequence this instantiation takes (score rows of PMAXT + 1 floats: 33 KB of LDS at 256, 98 KB at 768 —
// a loaded CU has the smaller hole far sooner)
template <int NK, bool EXACT, int PMAXT = ATTM_MAXP>  // MFMA k-steps covering the head dimension: 2*NK >= dk
__global__ __launch_bounds__(512) void attention_mfma_kernel(const float* qkv, long long bs, int ld, const int* len, int H,
                                                             int nheads, int window, const float* ek, const float* ev,
                                                             float* out, long long out_bs, int out_ld) {
  GLOW_PRIO();
  constexpr int PS = PMAXT + 1;  // odd row stride: conflict-free column reads
  __shared__ float S[32 * PS];
  __shared__ float vs[ATT_MAXDK * 65];  // V chunk [channel][64 keys]; afterwards the k-split partial outputs
  __shared__ float relS[32 * 33];
  __shared__ float evs[ATT_MAXW * ATT_MAXDK];
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int P = len[b];
  const int i0 = blockIdx.x * 32;
  if (i0 >= P) return;
  const int dk = EXACT ? 2 * NK : H / nheads;
  const int nrel = 2 * window + 1;
  const float scale = rsqrtf((float)dk);
  const float* q = qkv + (long long)b * bs + (long long)(h * dk) * ld;
  const float* k = q + (long long)H * ld;
  const float* v = k + (long long)H * ld;
  const int nkb = (P + 31) / 32;  // key blocks
  const int jpad = nkb * 32;
  const int col = lane & 31, half = lane >> 5;
  const int rbase = 4 * half;
  const int iq = min(i0 + col, P - 1);  // this lane's query column for A operands (clamped)

  // V chunk staging: thread t owns key column t & 63 of channels (t >> 6) + 8 u
  constexpr int VU = ATT_MAXDK / 8;
  const int vj = threadIdx.x & 63, vc0 = threadIdx.x >> 6;
  float tv[VU];
  auto v_fetch = [&](int j0) __attribute__((always_inline)) {
    const float* p = v + (long long)vc0 * ld + min(j0 + vj, P - 1);
#pragma unroll
    for (int u = 0; u < VU; ++u) {
      if (vc0 + 8 * u < dk) tv[u] = *p;  // wave-uniform
      p += 8 * ld;
    }
  };
  auto v_park = [&](int j0) __attribute__((always_inline)) {
    const bool ok = j0 + vj < P;
#pragma unroll
    for (int u = 0; u < VU; ++u)
      if (vc0 + 8 * u < dk) vs[(vc0 + 8 * u) * 65 + vj] = ok ? tv[u] : 0.f;
  };

  // ---- phase 1: S[i][j] = scale * q_i . k_j  (M = queries, N = keys, K-dim = channels).
  // One key block per wave per round; one extra block's "keys" are the 2w+1 relative-position
  // embeddings (attentions.py:228-234) — its columns >= 2w+1 hold junk that is never read.
  v_fetch(0);
  for (int e = threadIdx.x; e < nrel * ATT_MAXDK; e += 512) {
    const int rr = e >> 7, c = e & (ATT_MAXDK - 1);
    evs[e] = c < dk ? ev[rr * dk + c] : 0.f;  // consumed after several barriers
  }
  ATT_STAMP(1);
  if (wave <= nkb) {
    float av[NK];
    {
      // channel c = 2u + half; past dk (only when !EXACT) the pointer stops at the lane's last real channel and the
      // value is dropped by select: no branches, no second address stream
      const float* p = q + (long long)min(half, dk - 1) * ld + iq;
#pragma unroll
      for (int u = 0; u < NK; ++u) {
        const float t = *p;
        av[u] = (EXACT || 2 * u + half < dk) ? t : 0.f;
        p += (EXACT || 2 * u + 2 + half < dk) ? 2 * ld : 0;
      }
    }
    for (int nb = wave; nb <= nkb; nb += 8) {
      const bool band = nb == nkb;
      const int jk = min(nb * 32 + col, P - 1);
      const int rr = col < nrel ? col : nrel - 1;
      const int h0 = min(half, dk - 1);
      const float* p = band ? ek + rr * dk + h0 : k + (long long)h0 * ld + jk;
      const long long st = band ? 2 : 2 * (long long)ld;
      float bv[NK];
#pragma unroll
      for (int u = 0; u < NK; ++u) {
        const float t = *p;
        bv[u] = (EXACT || 2 * u + half < dk) ? t : 0.f;
        p += (EXACT || 2 * u + 2 + half < dk) ? st : 0;
      }
      att_floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int u = 0; u < NK; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      float* dst = band ? relS + rbase * 33 + col : S + rbase * PS + nb * 32 + col;
      const int rs = band ? 33 : PS;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * rs] = acc[r] * scale;  // keys >= P: never read
    }
  }
  ATT_STAMP(3);
  v_park(0);
  if (64 < jpad) v_fetch(64);  // in flight across the band add and the softmax
  __syncthreads();
  ATT_STAMP(4);
  // ---- relative-key band: S[i][i+r-w] += scale * q_i . Ek[r]
  for (int r = threadIdx.x >> 5; r < nrel; r += 16) {
    const int i = threadIdx.x & 31;
    const int gi = i0 + i;
    const int j = gi + r - window;
    if (gi < P && j >= 0 && j < P) S[i * PS + j] += relS[i * 33 + r];
  }
  __syncthreads();
  ATT_STAMP(5);
  // ---- softmax over keys: 16 lanes per row, the four rows of a wave at once (reductions are four xor steps);
  // keys P..jpad get the exact zero weight the reference's -1e4 fill underflows to.  Up to 256 keys a lane's
  // 16 scores stay in registers between the three passes.
  {
    const int sub = lane >> 4, l16 = lane & 15;
    float* row = S + (wave * 4 + sub) * PS;
    if (P <= 256) {
      float e[16];
      float mx = -3.0e38f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        e[t] = (l16 + 16 * t < P) ? row[l16 + 16 * t] : -3.0e38f;
        mx = fmaxf(mx, e[t]);
      }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float den = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        e[t] = (l16 + 16 * t < P) ? expf(e[t] - mx) : 0.f;
        den += e[t];
      }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) den += __shfl_xor(den, m);
      const float inv = 1.0f / den;
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (16 * t < jpad) row[l16 + 16 * t] = e[t] * inv;  // uniform; jpad is a multiple of 32
    } else {
      float mx = -3.0e38f;
      for (int j = l16; j < P; j += 16) mx = fmaxf(mx, row[j]);
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
      float den = 0.f;
      for (int j = l16; j < P; j += 16) {
        const float e = expf(row[j] - mx);
        row[j] = e;
        den += e;
      }
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) den += __shfl_xor(den, m);
      const float inv = 1.0f / den;
      for (int j = l16; j < jpad; j += 16) row[j] = (j < P) ? row[j] * inv : 0.f;
    }
  }
  __syncthreads();
  ATT_STAMP(6);
  // ---- phase 2: O^T[c][i] = sum_j v[c][j] p[i][j]   (M = channels, N = queries, K-dim = keys).  A wave is one
  // (32-channel block, key slice) unit: the 8-key groups of a chunk are dealt round-robin to the nks slices.
  const int ncb = (dk + 31) / 32;           // <= 4
  const int nks = ncb == 3 ? 2 : 8 / ncb;   // 8, 4, 2, 2 slices: ncb * nks <= 8 waves
  const int cb = wave % ncb, ks = wave / ncb;
  const bool unit = ks < nks;
  att_floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int j0 = 0; j0 < jpad; j0 += 64) {
    if (unit) {
      const int cl = cb * 32 + col;
      const float* va = vs + min(cl, dk - 1) * 65 + half;
      const float* pb = S + col * PS + j0 + half;
      const int jn = min(64, jpad - j0);
      for (int jj = 8 * ks; jj < jn; jj += 8 * nks) {  // jn is a multiple of 32
        float a4[4], b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a4[u] = (EXACT || cl < dk) ? va[jj + 2 * u] : 0.f;
          b4[u] = pb[jj + 2 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u], b4[u], acc, 0, 0, 0);
      }
    }
    if (j0 + 64 < jpad) {  // uniform over the workgroup
      __syncthreads();
      v_park(j0 + 64);
      if (j0 + 128 < jpad) v_fetch(j0 + 128);
      __syncthreads();
    }
  }
  // the relative values on the band (attentions.py:246-253) are one more K-block of the same GEMM:
  // O^T[c][i] += sum_r Ev[r][c] p[i][i+r-w]; the last slice, which has the fewest key groups, takes it
  if (unit && ks == nks - 1) {
    const int cl = cb * 32 + col;
    const int gi = i0 + col;
    for (int u = 0; 2 * u < nrel; ++u) {
      const int rr = 2 * u + half;
      const int j = gi + rr - window;
      const float a = (rr < nrel && cl < dk) ? evs[rr * ATT_MAXDK + cl] : 0.f;
      const float p = (rr < nrel && j >= 0 && j < P && gi < P) ? S[col * PS + j] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p, acc, 0, 0, 0);
    }
  }
  ATT_STAMP(7);
  // ---- the slices' partial tiles go through LDS as [slice][channel][query] (over the dead V chunk) ...
  __syncthreads();
  float* red = vs;  // nks * ncb * 32 * 32 floats <= 8192 <= ATT_MAXDK * 65
  if (unit) {
    float* dst = red + ((ks * ncb + cb) * 32 + rbase) * 32 + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * 32] = acc[r];
  }
  __syncthreads();
  // ---- ... and thread t sums and stores channels (t >> 5) + 16 m of query t & 31, lanes along time
  {
    const int i = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const int gi = i0 + i;
    if (gi >= P) return;
    float* ob = out + (long long)b * out_bs + (long long)(h * dk) * out_ld + gi;
    for (int c = cg; c < dk; c += 16) {
      const int blk = c >> 5, cl = c & 31;
      float o = 0.f;
      for (int s = 0; s < nks; ++s) o += red[((s * ncb + blk) * 32 + cl) * 32 + i];
      ob[(long long)c * out_ld] = o;
    }
  }
  ATT_STAMP(8);
}

// G9a: durations.  w = exp(logw)*length_scale, w_ceil = ceil(w); cum = inclusive
// cumsum; frames = max(sum,1) truncated to a multiple of n_sqz
// (glow_tts/models.py:323-336).  One workgroup per batch row.
__global__ void duration_kernel(const float* logw, long long bs, const int* len, float length_scale, int n_sqz,
                                int* cum, int cum_ld, int* frames, int max_frames_cap) {
  GLOW_PRIO();
  // one wave per row: lane l owns the contiguous run [l*per, (l+1)*per) of ids, sums its
  // durations, the 64 run totals are scanned with __shfl_up, and the run is walked a second
  // time to write the inclusive cumsum.  Durations are integers (ceil), so the int sum equals
  // the reference's float cumsum exactly; one duration is capped at max_frames_cap (exp overflow).
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  const int P = len[b];
  const int per = (P + 63) / 64;
  const int t0 = lane * per;
  const int t1 = t0 + per < P ? t0 + per : P;
  const float* lw = logw + (long long)b * bs;
  const float capf = (float)max_frames_cap;  // <= 2^28: two capped terms still fit an int
  auto dur = [&](int t) { return (int)fminf(fmaxf(ceilf(expf(lw[t]) * length_scale), 0.f), capf); };
  // the scan runs UNCLAMPED in 64 bits (exact for any input: <= 2^28 per id) and only what is written is capped, so
  // cum[] stays monotone also when the sum overflows the cap (the call then fails with NOMEM on the host side)
  long long run = 0;
  for (int t = t0; t < t1; ++t) run += dur(t);
  long long incl = run;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned lo = __shfl_up((unsigned)(incl & 0xffffffffLL), d), hi = __shfl_up((unsigned)(incl >> 32), d);
    const long long o = ((long long)hi << 32) | (long long)lo;
    if (lane >= d) incl += o;
  }
  long long acc = incl - run;  // exclusive prefix of this lane's run
  for (int t = t0; t < t1; ++t) {
    acc += dur(t);
    cum[(long long)b * cum_ld + t] = (int)(acc < (long long)max_frames_cap ? acc : (long long)max_frames_cap);
  }
  const long long total64 = ((long long)__shfl((unsigned)(incl >> 32), 63) << 32) | (long long)__shfl((unsigned)(incl & 0xffffffffLL), 63);
  const int total = (int)(total64 < (long long)max_frames_cap ? total64 : (long long)max_frames_cap);
  if (lane == 0) {
    int y = total > 1 ? total : 1;
    y = (y / n_sqz) * n_sqz;
    if (y > max_frames_cap) y = (max_frames_cap / n_sqz) * n_sqz;
    frames[b] = y;
  }
}

__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
// counter-based N(0,1): two hashed uniforms -> Box-Muller
__device__ __forceinline__ float gauss_noise(uint64_t seed, uint32_t b, uint32_t c, uint32_t t) {
  uint32_t k = pcg_hash((uint32_t)seed ^ pcg_hash((uint32_t)(seed >> 32) + 0x9e3779b9u));
  uint32_t x = pcg_hash(k ^ pcg_hash(b * 0x85ebca6bu + c) ^ (t * 0xc2b2ae35u));
  uint32_t y = pcg_hash(x ^ 0x27d4eb2fu);
  const float u1 = ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(y >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795864f * u2);
}

// The generator alone, out[b][c][t] = the draw of row stream seed + b (as expand_noise_squeeze_kernel keys it): what the
// distribution tests look at.
__global__ void noise_fill_kernel(float* out, int C, int T, uint64_t seed) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T) out[((long long)b * C + c) * T + t] = gauss_noise(seed + (uint64_t)b, 0u, (uint32_t)c, (uint32_t)t);
}

// G9b+G10+G11: frame j of row b repeats id idx(j) = #{t : cum[t] <= j}
// (glow_tts/utils.py:99-115 `generate_path`, never materialised as a matrix);
// z = x_m[:, idx] + noise * noise_scale (models.py:348), written directly in the
// squeezed layout z_sqz[s*M + c][j/n] (utils.py:135-147).
__global__ void expand_noise_squeeze_kernel(const float* xm, long long xm_bs, int xm_ld, const int* len,
                                            const int* cum, int cum_ld, const int* frames, const float* noise,
                                            long long noise_bs, int noise_ld, float noise_scale, uint64_t seed,
                                            const unsigned long long* row_seeds, int M, int n_sqz, float* z, long long z_bs, int z_ld) {
  GLOW_PRIO();
  const int b = blockIdx.z;
  // the noise stream of a row is keyed by the ROW's seed (row_seeds[b], default seed + b) and nothing else: an
  // utterance draws the same field in a batch as in a call of its own with that seed
  const uint64_t rseed = row_seeds ? (uint64_t)row_seeds[b] : seed + (uint64_t)b;
  const int F = frames[b];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= F) return;
  const int P = len[b];
  const int* cb = cum + (long long)b * cum_ld;
  int lo = 0, hi = P;  // first t with cum[t] > j
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cb[mid] <= j) lo = mid + 1; else hi = mid;
  }
  const int id = lo < P ? lo : P - 1;
  const int s = j % n_sqz, j2 = j / n_sqz;
  for (int c = blockIdx.y; c < M; c += gridDim.y) {
    float v = xm[(long long)b * xm_bs + (long long)c * xm_ld + id];
    if (noise_scale != 0.f) {
      const float nz = noise ? noise[(long long)b * noise_bs + (long long)c * noise_ld + j]
                             : gauss_noise(rseed, 0u, (uint32_t)c, (uint32_t)j);
      v += nz * noise_scale;
    }
    z[(long long)b * z_bs + (long long)(s * M + c) * z_ld + j2] = v;
  }
}

// G12b+G12a: InvConvNear reverse (pre-inverted 4x4, glow_tts/layers.py:238-272)
// followed by ActNorm reverse (layers.py:192-194), in place on x[B][C][ld].
// Group k mixes channels {a*C/2 + k*(ns/2) + s}, n = a*(ns/2)+s.
__global__ void invconv_actnorm_kernel(float* x, long long bs, int ld, const int* frames, int div, int C, int ns,
                                       const float* winv, const float* an_bias, const float* an_scale) {
  const int b = blockIdx.z;
  const int T = frames[b] / div;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int groups = C / ns;
  const int hs = ns / 2;
  float* xb = x + (long long)b * bs + t;
  for (int k = blockIdx.y; k < groups; k += gridDim.y) {
    float in[8], o[8];
    for (int n = 0; n < ns; ++n) {
      const int a = n / hs, s = n - a * hs;
      in[n] = xb[(long long)(a * (C / 2) + k * hs + s) * ld];
    }
    for (int m = 0; m < ns; ++m) {
      float acc = 0.f;
      for (int n = 0; n < ns; ++n) acc += winv[m * ns + n] * in[n];
      o[m] = acc;
    }
    for (int m = 0; m < ns; ++m) {
      const int a = m / hs, s = m - a * hs;
      const int c = a * (C / 2) + k * hs + s;
      xb[(long long)c * ld] = (o[m] - an_bias[c]) * an_scale[c];
    }
  }
}

struct MelTransform {
  int signal_norm, symmetric_norm, clip_norm, convert_db_to_amp, do_drc;
  float min_level_db, max_norm, ref_level_db, spec_gain;
};

// The numpy mel transforms of `_sentence_task` (larynx/__init__.py:242-249 ->
// larynx/audio.py:83-108), kept in the reference's pow -> log form because of
// the 1e-5 clamp.
__device__ __forceinline__ float mel_transform(float v, const MelTransform& m) {
  if (m.signal_norm) {
    if (m.symmetric_norm) {
      if (m.clip_norm) v = fminf(fmaxf(v, -m.max_norm), m.max_norm);
      v = ((v + m.max_norm) * -m.min_level_db / (2.0f * m.max_norm)) + m.min_level_db;
    } else {
      if (m.clip_norm) v = fminf(fmaxf(v, 0.f), m.max_norm);
      v = (v * -m.min_level_db / m.max_norm) + m.min_level_db;
    }
    v += m.ref_level_db;
  }
  if (m.convert_db_to_amp) v = powf(10.0f, v / m.spec_gain);
  if (m.do_drc) v = logf(fmaxf(v, 1e-5f));
  return v;
}

// G13 + M1: unsqueeze [C][F/n] -> [M][F] (glow_tts/utils.py:150-160), write the
// raw mel (the GlowTTS output the parity check is defined on) and the vocoder
// input; the padded tail [F_b, ld) of both is zero-filled.
__global__ void mel_finalize_kernel(const float* x, long long x_bs, int x_ld, const int* frames, int M, int n_sqz,
                                    float* mel, float* mel_voc, long long mel_bs, int mel_ld, MelTransform mt,
                                    int apply) {
  GLOW_PRIO();
  const int b = blockIdx.z;
  const int F = frames[b];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= mel_ld) return;
  const int s = j % n_sqz, j2 = j / n_sqz;
  for (int c = blockIdx.y; c < M; c += gridDim.y) {
    float v = 0.f, u = 0.f;
    if (j < F) {
      v = x[(long long)b * x_bs + (long long)(s * M + c) * x_ld + j2];
      u = apply ? mel_transform(v, mt) : v;
    }
    const long long o = (long long)b * mel_bs + (long long)c * mel_ld + j;
    mel[o] = v;
    mel_voc[o] = u;
  }
}

// M1 alone, for mels that arrive from the host (drop-in `mels_to_audio`).
__global__ void mel_transform_kernel(const float* in, float* out, long long n, MelTransform mt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mel_transform(in[i], mt);
}

// H5 pass 1: per-row max |a| as uint bits (monotone for non-negative floats).
__global__ void absmax_kernel(const float* wav, long long bs, const int* frames, int hop, unsigned* peak_bits) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const long long N = (long long)frames[b] * hop;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(wav[(long long)b * bs + i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = red[0];
    for (unsigned w = 1; w < (blockDim.x + 63) / 64; ++w) r = fmaxf(r, red[w]);
    atomicMax(&peak_bits[b], __float_as_uint(r));
  }
}

// H5 pass 2: `audio_float_to_int16` (larynx/audio.py:118-125): a * 32767/max(0.01,peak),
// clip, truncate toward zero; the padded tail is zero.
__global__ void to_int16_kernel(const float* wav, long long bs, const int* frames, int hop, const unsigned* peak_bits,
                                short* out, long long out_bs, long long out_ld, int pad_before) {
  // `pad_before` zero samples precede the audio in the output row (SSML <break> before the
  // sentence, larynx/__init__.py:277-283); everything past the audio up to out_ld is zero
  // too, which covers the pause after the sentence.
  const int b = blockIdx.y;
  const long long N = (long long)frames[b] * hop;
  const float peak = fmaxf(0.01f, __uint_as_float(peak_bits[b]));
  const float g = 32767.0f / peak;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < out_ld; i += (long long)gridDim.x * blockDim.x) {
    short s = 0;
    const long long j = i - pad_before;
    if (j >= 0 && j < N) {
      float v = wav[(long long)b * bs + j] * g;
      v = fminf(fmaxf(v, -32767.0f), 32767.0f);
      s = (short)(int)v;
    }
    out[(long long)b * out_bs + i] = s;
  }
}

// ---- spectral-subtraction denoiser -------------------
// `HiFiGanVocoder.denoise` (larynx/hifi_gan.py:171-179) with the reference's own
// STFT conventions (larynx/audio.py:232-269): 1024-point frames every 256 samples
// for i in range(0, N - 1024, 256), symmetric np.hanning window on analysis AND
// synthesis, no window-sum normalisation.  One workgroup = one frame: radix-2
// FFT in LDS, |X| -= bias*strength (clamped at 0, phase kept), inverse FFT,
// synthesis window; a second kernel overlap-adds the frames in frame order.
constexpr int DN_FFT = 1024;
constexpr int DN_HOP = 256;

__device__ __forceinline__ int dn_brev10(int i) {
  int r = 0;
#pragma unroll
  for (int b = 0; b < 10; ++b) r |= ((i >> b) & 1) << (9 - b);
  return r;
}

__device__ __forceinline__ void dn_fft1024(float* re, float* im, const float* cs, const float* sn, int tid) {
  // in-place radix-2 DIT on bit-reversed input; 512 butterflies per stage, 2 per thread
  for (int s = 1; s <= 10; ++s) {
    const int half = 1 << (s - 1);
    const int stride = DN_FFT >> s;  // twiddle index step
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = tid + 256 * u;
      const int pos = t & (half - 1);
      const int i0 = ((t - pos) << 1) + pos;
      const int i1 = i0 + half;
      const float wr = cs[pos * stride], wi = sn[pos * stride];
      const float tr = re[i1] * wr - im[i1] * wi;
      const float ti = re[i1] * wi + im[i1] * wr;
      const float ar = re[i0], ai = im[i0];
      re[i1] = ar - tr;
      im[i1] = ai - ti;
      re[i0] = ar + tr;
      im[i0] = ai + ti;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int dn_num_frames(long long N) { return N > DN_FFT ? (int)((N - DN_FFT + DN_HOP - 1) / DN_HOP) : 0; }

// mag_out != nullptr: only write |X| of frame 0 (bias spectrum initialisation).
__global__ __launch_bounds__(256) void stft_denoise_kernel(const float* wav, long long bs, const int* frames, int hop,
                                                           const float* bias_spec, float strength, float* fbuf, int Tmax,
                                                           float* mag_out) {
  __shared__ float re[DN_FFT], im[DN_FFT], cs[DN_FFT / 2], sn[DN_FFT / 2], win[DN_FFT];
  const int b = blockIdx.y, n = blockIdx.x, tid = threadIdx.x;
  const long long N = (long long)frames[b] * hop;
  if (n >= dn_num_frames(N)) return;
  const float* x = wav + (long long)b * bs + (long long)n * DN_HOP;
  for (int i = tid; i < DN_FFT; i += 256) {
    const float w = 0.5f - 0.5f * cospif(2.0f * (float)i / (float)(DN_FFT - 1));  // np.hanning(1024)
    win[i] = w;
    const int j = dn_brev10(i);
    re[j] = x[i] * w;
    im[j] = 0.f;
  }
  for (int k = tid; k < DN_FFT / 2; k += 256) {
    cs[k] = cospif((float)k / 512.0f);
    sn[k] = -sinpif((float)k / 512.0f);
  }
  __syncthreads();
  dn_fft1024(re, im, cs, sn, tid);
  if (mag_out) {
    for (int k = tid; k <= DN_FFT / 2; k += 256) mag_out[k] = sqrtf(re[k] * re[k] + im[k] * im[k]);
    return;
  }
  // spectral subtraction on bins 0..512, mirrored onto the conjugate half; conj for the inverse
  for (int k = tid; k <= DN_FFT / 2; k += 256) {
    const float mag = sqrtf(re[k] * re[k] + im[k] * im[k]);
    const float g = mag > 0.f ? fmaxf(mag - bias_spec[k] * strength, 0.f) / mag : 0.f;
    const float xr = re[k] * g, xi = im[k] * g;
    re[k] = xr;
    im[k] = -xi;  // conj(X)
    if (k > 0 && k < DN_FFT / 2) {
      re[DN_FFT - k] = xr;
      im[DN_FFT - k] = xi;  // conj of the mirrored bin conj(X[k])
    }
  }
  __syncthreads();
  // bit-reverse permutation in place, then the same forward transform: IFFT(X) = conj(FFT(conj X))/N
  for (int i = tid; i < DN_FFT; i += 256) {
    const int j = dn_brev10(i);
    if (i < j) {
      const float tr = re[i], ti = im[i];
      re[i] = re[j];
      im[i] = im[j];
      re[j] = tr;
      im[j] = ti;
    }
  }
  __syncthreads();
  dn_fft1024(re, im, cs, sn, tid);
  float* fo = fbuf + ((long long)b * Tmax + n) * DN_FFT;
  for (int i = tid; i < DN_FFT; i += 256) fo[i] = win[i] * re[i] * (1.0f / DN_FFT);
}

// out[s] = sum over the (up to 4) frames covering sample s, in frame order; length T*256 + 1024.
__global__ void overlap_add_kernel(const float* fbuf, int Tmax, const int* frames, int hop, float* out, long long bs,
                                   long long ld) {
  const int b = blockIdx.y;
  const long long N = (long long)frames[b] * hop;
  const int T = dn_num_frames(N);
  const long long len = T > 0 ? (long long)T * DN_HOP + DN_FFT : 0;
  for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < ld; s += (long long)gridDim.x * blockDim.x) {
    float acc = 0.f;
    if (s < len) {
      int n0 = (int)((s - (DN_FFT - 1) + DN_HOP - 1) / DN_HOP);
      if (s < DN_FFT) n0 = 0;
      int n1 = (int)(s / DN_HOP);
      if (n1 > T - 1) n1 = T - 1;
      for (int n = n0; n <= n1; ++n) acc += fbuf[((long long)b * Tmax + n) * DN_FFT + (s - (long long)n * DN_HOP)];
    }
    out[(long long)b * bs + s] = acc;
  }
}

__global__ void zero_tail_kernel(float* wav, long long bs, long long ld, const int* frames, int hop) {
  const int b = blockIdx.y;
  const long long N = (long long)frames[b] * hop;
  for (long long i = N + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < ld; i += (long long)gridDim.x * blockDim.x)
    wav[(long long)b * bs + i] = 0.f;
}

__global__ void fill_int_kernel(int* p, int n, int v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-speaker voices (glow_tts/models.py:304-306, 318-319).  g = F.normalize(emb_g(speaker)) is a [gin] vector per batch
// row; everything the reference computes from it is a matrix-vector product, done here once per call:
//   * workgroups x < nblk: cond[b][blk][:] = cond_layer_blk(g) (layers.py:109-113, 141-142 — a 1x1 conv of a length-1 tensor),
//     the [2H * n_layers] gate offsets of flow block blk; a layer's gate conv adds its [2H] slice (conv_mfma.h EPI_GATE,
//     gate16.h);
//   * workgroup x = nblk: dps[b][co][k] = sum_ci W1[co][H + ci][k] g[ci], the speaker half of the duration predictor's first
//     conv (models.py:114-116, 128-132: g is repeated along time and concatenated to the encoder output) per tap —
//     speaker_dp_plane_kernel turns it into the [Fd][P] plane the conv of the encoder half takes as its residual.
constexpr int SPEAKER_MAX_GIN = 1024;
__global__ __launch_bounds__(256) void speaker_cond_kernel(const float* __restrict__ emb, int n_spk, int gin, const int* __restrict__ spk,
                                                           const float* __restrict__ cw, const float* __restrict__ cb, int nblk, int n2,
                                                           float* __restrict__ cond, const float* __restrict__ dpw, int FdK, int K,
                                                           float* __restrict__ dps) {
  GLOW_PRIO();
  __shared__ float g[SPEAKER_MAX_GIN];
  __shared__ float red[256];
  const int tid = threadIdx.x, b = blockIdx.y;
  int sid = spk[b];
  sid = sid < 0 ? 0 : (sid >= n_spk ? n_spk - 1 : sid);  // (host ids are range-checked before the call)
  const float* e = emb + (size_t)sid * gin;
  float ss = 0.f;
  for (int i = tid; i < gin; i += 256) {
    const float v = e[i];
    g[i] = v;
    ss += v * v;
  }
  red[tid] = ss;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const float nrm = sqrtf(red[0]);
  const float inv = 1.0f / (nrm > 1e-12f ? nrm : 1e-12f);  // F.normalize: x / max(||x||_2, eps)
  for (int i = tid; i < gin; i += 256) g[i] *= inv;
  __syncthreads();
  if ((int)blockIdx.x < nblk) {
    const int blk = blockIdx.x;
    for (int o = tid; o < n2; o += 256) {
      const float* w = cw + ((size_t)blk * n2 + o) * gin;
      float acc = 0.f;
      for (int i = 0; i < gin; ++i) acc += w[i] * g[i];
      cond[((size_t)b * nblk + blk) * n2 + o] = acc + cb[(size_t)blk * n2 + o];
    }
  } else {
    for (int o = tid; o < FdK; o += 256) {
      const int co = o / K, kk = o - co * K;
      const float* w = dpw + (size_t)co * gin * K + kk;
      float acc = 0.f;
      for (int i = 0; i < gin; ++i) acc += w[(size_t)i * K] * g[i];
      dps[(size_t)b * FdK + o] = acc;
    }
  }
}
// gc[b][co][t] = sum over the taps k whose input position t + k - pad lies inside the row (the reference masks the
// concatenated input before the conv, models.py:42: zero padding AND zeros past the row's length)
__global__ __launch_bounds__(256) void speaker_dp_plane_kernel(const float* __restrict__ dps, int Fd, int K, int pad, const int* __restrict__ len,
                                                               float* __restrict__ gc, long long bs, int ld) {
  GLOW_PRIO();
  const int b = blockIdx.z, co = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ld) return;
  const int L = len[b];
  float v = 0.f;
  if (t < L) {
    const float* s = dps + ((size_t)b * Fd + co) * K;
    for (int k = 0; k < K; ++k) {
      const int tt = t + k - pad;
      if (tt >= 0 && tt < L) v += s[k];
    }
  }
  gc[(long long)b * bs + (long long)co * ld + t] = v;
}

}  // namespace mi355tts
