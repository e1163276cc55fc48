// The vocoder's output tail: conv_post + tanh (+ the peak for the int16 scaling) and the delivery of the finished rows
// (hifi_gan/models.py:198-201; larynx/audio.py:118-125 audio_float_to_int16; the SSML pauses of larynx/__init__.py:277-283).
//
//   post_conv_kernel — x = tanh(conv_post(leaky_relu(x, 0.01))): C -> ONE output row, k = 7.  On the generic MFMA tile that
//     row is padded to 32 (1/32 of the matrix work is real) and the launch is bound by its staging: 41 us for 70 MFLOP at the
//     standard utterance.  Here: a [16 channels x 264 columns] tile of the averaged, activated input in LDS, every thread four
//     consecutive outputs of four channels as a sliding window over three aligned float4 reads per channel (VALU fmaf chains;
//     HBM/L2-bound: 3 x 20 MB of input planes per launch), the four channel groups summed through LDS, tanh, one float4 store —
//     and the |max| of the workgroup's columns in the same pass (the row's peak for the int16 scaling was a launch of its own).
//   wave_out_kernel — one launch writes what the caller gets: the f32 row (pad_before zeros | samples | zeros up to its
//     stride) and / or the int16 row (same layout, audio_float_to_int16's scaling and truncation), replacing zero_tail +
//     to_int16 + up to five memset / memcpy launches per row.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355tts {

constexpr int POST_TW = 256;          // output columns per workgroup
constexpr int POST_XW = POST_TW + 8;  // staged columns (4 of halo on either side)
constexpr int POST_CC = 16;           // channels per staged chunk

struct PostArgs {
  const float* x;   // [B][C][ld]; with x2 (and x3): the stage input is (x + x2 [+ x3]) / in_div — the MRF average on load
  const float* x2;
  const float* x3;
  float in_div, slope;
  long long x_bs;
  int x_ld;
  const int* len;  // valid columns per batch row: len ? len[b] * len_mul : len_const
  int len_mul, len_const;
  const float* w;  // [C][K] (the checkpoint's conv_post.weight[0])
  const float* bias;
  int C;
  float* y;  // [B][y_bs]
  long long y_bs;
  // max |y| of every workgroup's 256 columns at peak[b * peak_ld + blockIdx.x] (no atomics, nothing to zero beforehand:
  // wave_out_kernel takes the maximum over a row's ceil(L / 256) entries); nullptr: not wanted
  float* peak;
  long long peak_ld;
};

// NPL = input planes (1: x; 2 / 3: the MRF average (x + x2 [+ x3]) / in_div taken on load).  A template parameter, not a test
// of the pointers: a load behind a (wave-uniform) branch makes the compiler drain vmcnt at the join, one memory round trip per
// plane; here all planes of a chunk are requested back to back, and the next chunk's while this one is computed.
template <int K, int NPL>
__global__ __launch_bounds__(256) void post_conv_kernel(const PostArgs a) {
  static_assert(K == 7, "halo of 4 columns either side");
  __shared__ float xs[POST_CC * POST_XW];
  __shared__ float red[3 * POST_TW];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int t0 = blockIdx.x * POST_TW;
  if (t0 >= L) return;
  const int i = tid & 63, h = tid >> 6;  // wave h: outputs t0 + 4 i ... + 3, channels 4 h ... 4 h + 3 of each chunk
  constexpr int XW4 = POST_XW / 4, NF4 = POST_CC * XW4, NE = (NF4 + 255) / 256;
  const float* xb[3] = {a.x + (long long)b * a.x_bs, (NPL > 1 ? a.x2 : a.x) + (long long)b * a.x_bs,
                        (NPL > 2 ? a.x3 : a.x) + (long long)b * a.x_bs};
  float4 pre[NPL][NE];
  auto request = [&](int c0) {  // 16-byte loads at clamped addresses
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int f4 = tid + 256 * e;
      const int row = f4 / XW4 < POST_CC ? f4 / XW4 : POST_CC - 1, f = f4 - (f4 / XW4) * XW4;
      const int ch = c0 + row < a.C ? c0 + row : a.C - 1;
      const int col = t0 - 4 + 4 * f;
      const long long off = (long long)ch * a.x_ld + (col < 0 ? 0 : (col > a.x_ld - 4 ? a.x_ld - 4 : col));
#pragma unroll
      for (int p = 0; p < NPL; ++p) pre[p][e] = *reinterpret_cast<const float4*>(xb[p] + off);
    }
  };
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  request(0);
  for (int c0 = 0; c0 < a.C; c0 += POST_CC) {
    if (c0) __syncthreads();
    // ---- the chunk's tile: the planes' sum, / in_div, zero outside [0, L), leaky ReLU
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int f4 = tid + 256 * e;
      const int row = f4 / XW4, f = f4 - row * XW4;
      const int col = t0 - 4 + 4 * f;
      const bool rok = c0 + row < a.C;
      float4 v = pre[0][e];
      if constexpr (NPL > 1) {
        v.x += pre[1][e].x;
        v.y += pre[1][e].y;
        v.z += pre[1][e].z;
        v.w += pre[1][e].w;
        if constexpr (NPL > 2) {
          v.x += pre[2][e].x;
          v.y += pre[2][e].y;
          v.z += pre[2][e].z;
          v.w += pre[2][e].w;
        }
        v.x = v.x / a.in_div;
        v.y = v.y / a.in_div;
        v.z = v.z / a.in_div;
        v.w = v.w / a.in_div;
      }
      v.x = (rok && col >= 0 && col < L) ? v.x : 0.f;
      v.y = (rok && col + 1 >= 0 && col + 1 < L) ? v.y : 0.f;
      v.z = (rok && col + 2 >= 0 && col + 2 < L) ? v.z : 0.f;
      v.w = (rok && col + 3 >= 0 && col + 3 < L) ? v.w : 0.f;
      v.x = v.x > 0.f ? v.x : v.x * a.slope;
      v.y = v.y > 0.f ? v.y : v.y * a.slope;
      v.z = v.z > 0.f ? v.z : v.z * a.slope;
      v.w = v.w > 0.f ? v.w : v.w * a.slope;
      if (f4 < NF4) reinterpret_cast<float4*>(xs)[f4] = v;
    }
    __syncthreads();
    request(c0 + POST_CC < a.C ? c0 + POST_CC : c0);  // (past the last chunk: re-reads it, L2 hits, nobody waits for them)
    // ---- four channels of this chunk: output t0 + 4 i + e, tap k reads staged column 4 i + e + k + 1
#pragma unroll
    for (int cc = 0; cc < POST_CC / 4; ++cc) {
      const int c = c0 + 4 * h + cc;  // wave-uniform
      const float4* xr = reinterpret_cast<const float4*>(xs + (4 * h + cc) * POST_XW) + i;
      const float4 p0 = xr[0], p1 = xr[1], p2 = xr[2];
      const float win[12] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w, p2.x, p2.y, p2.z, p2.w};
      const float* wc = a.w + (c < a.C ? c : a.C - 1) * K;
      const float live = c < a.C ? 1.0f : 0.0f;  // (rows past C are staged as zeros as well)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float wk = wc[k] * live;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(wk, win[e + k + 1], acc[e]);
      }
    }
  }
  // ---- the four channel groups meet in LDS (((g0 + g1) + g2) + g3), bias, tanh, store, peak
  if (h > 0) *reinterpret_cast<float4*>(red + (h - 1) * POST_TW + 4 * i) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  float m = 0.f;
  if (h == 0) {
    const float bs = a.bias ? a.bias[0] : 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4 o = *reinterpret_cast<const float4*>(red + g * POST_TW + 4 * i);
      acc[0] += o.x;
      acc[1] += o.y;
      acc[2] += o.z;
      acc[3] += o.w;
    }
    float4 v;
    v.x = tanhf(acc[0] + bs);
    v.y = tanhf(acc[1] + bs);
    v.z = tanhf(acc[2] + bs);
    v.w = tanhf(acc[3] + bs);
    const int t = t0 + 4 * i;
    float* yp = a.y + (long long)b * a.y_bs + t;
    if (t + 3 < L) {
      *reinterpret_cast<float4*>(yp) = v;
      m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
      if (t < L) { yp[0] = v.x; m = fmaxf(m, fabsf(v.x)); }
      if (t + 1 < L) { yp[1] = v.y; m = fmaxf(m, fabsf(v.y)); }
      if (t + 2 < L) { yp[2] = v.z; m = fmaxf(m, fabsf(v.z)); }
    }
  }
  if (a.peak && h == 0) {  // wave 0 holds every output of the tile
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if (tid == 0) a.peak[(long long)b * a.peak_ld + blockIdx.x] = m;
  }
}

constexpr int VOC_MAX_ROWS = 16;  // rows of a call with per-row destinations

struct WaveOutArgs {
  const float* wav;  // [B][bs] finished float rows
  long long bs;
  const int* frames;
  int hop;
  // int16 output only: row b's peak = max over peak[b * peak_ld + 0 .. npeak) where npeak = peak_parts ? peak_parts :
  // ceil(samples / 256) (post_conv_kernel's per-workgroup maxima; 1 = one value per row, e.g. behind the denoiser)
  const float* peak;
  long long peak_ld;
  int peak_parts;
  float* f32;            // optional: [B][f_bs], row = pad_before zeros | samples | zeros up to f_ld
  long long f_bs, f_ld;
  short* i16;  // optional: same layout
  long long i_bs, i_ld;
  int pad_before;
  // per_row != 0: the rows of a coalesced call (host_join.h) go to different callers — row b's float / int16 destination, row
  // length and pause come from the tables below instead of base + b * stride (a null entry = that row has no such output)
  int per_row;
  float* f32_rows[VOC_MAX_ROWS];
  short* i16_rows[VOC_MAX_ROWS];
  long long f_ld_rows[VOC_MAX_ROWS], i_ld_rows[VOC_MAX_ROWS];
  int pad_rows[VOC_MAX_ROWS];
};

__global__ __launch_bounds__(256) void wave_out_kernel(const WaveOutArgs a) {
  const int b = blockIdx.y;
  const long long N = (long long)a.frames[b] * a.hop;
  const float* src = a.wav + (long long)b * a.bs;
  const long long step = (long long)gridDim.x * blockDim.x, first = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int pad_before = a.per_row ? a.pad_rows[b] : a.pad_before;
  float* const f32 = a.per_row ? a.f32_rows[b] : (a.f32 ? a.f32 + (long long)b * a.f_bs : nullptr);
  short* const i16 = a.per_row ? a.i16_rows[b] : (a.i16 ? a.i16 + (long long)b * a.i_bs : nullptr);
  if (f32) {
    float* dst = f32;
    const long long f_ld = a.per_row ? a.f_ld_rows[b] : a.f_ld;
    for (long long i = first; i < f_ld; i += step) {
      const long long j = i - pad_before;
      dst[i] = (j >= 0 && j < N) ? src[j] : 0.f;
    }
  }
  if (i16) {  // (uniform per workgroup: the barrier below is safe)
    __shared__ float pm[4];
    const int np = a.peak_parts ? a.peak_parts : (int)((N + POST_TW - 1) / POST_TW);
    float m = 0.f;
    for (int i = threadIdx.x; i < np; i += blockDim.x) m = fmaxf(m, a.peak[(long long)b * a.peak_ld + i]);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft));
    if ((threadIdx.x & 63) == 0) pm[threadIdx.x >> 6] = m;
    __syncthreads();
    const float peak = fmaxf(0.01f, fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3])));
    const float g = 32767.0f / peak;
    short* dst = i16;
    const long long i_ld = a.per_row ? a.i_ld_rows[b] : a.i_ld;
    for (long long i = first; i < i_ld; i += step) {
      short s = 0;
      const long long j = i - pad_before;
      if (j >= 0 && j < N) {
        float v = src[j] * g;
        v = fminf(fmaxf(v, -32767.0f), 32767.0f);
        s = (short)(int)v;
      }
      dst[i] = s;
    }
  }
}

}  // namespace mi355tts
