// The WaveNet of a GlowTTS coupling block (glow_tts/layers.py:138-162, reverse pass of attentions.py:119-142) in fp16 as ONE
// launch — the acoustic model's share of the reference's `half` switch (`.half()` on the FlowGenerator, larynx/glow_tts.py:90-91).
//
// In f32 a block's WN is 4 gate-conv launches + 3 res_skip launches of ~240 short workgroups each: latency chains that hold the
// whole chip for 5-8 us apiece (a column-owner form lost in f32 because ONE workgroup then carries 6912 f32
// MFMAs per layer).  The fp16 matrix rate is 16 x the f32 one, so here a workgroup DOES own its columns: it keeps the hidden
// state h [H x 64 columns] and the gated activations in LDS as fp16 octet rows (conv_f16.h's layout: the B operand as it stands),
// runs all layers back to back — gate conv (k taps, H -> 2H) -> tanh * sigmoid -> res_skip 1 x 1 -> h += res, skip += skp — and
// hands the LAST layer's gated activations and the skip sum of the earlier layers to glow_tail_kernel (f32, unchanged) in the
// planes the f32 chain would have left them in.  A layer's conv reaches (k - 1) / 2 columns to either side, so of a tile's 64
// columns 64 - (k - 1) n_layers are exact (48 at k = 5, 4 layers): tiles advance by that much and recompute the margins.
// 7 workgroups per block at the standard utterance instead of ~1700 over 7 launches; 2 launches per block instead of 8.
//
// Rounding: weights, h and the gated activations are fp16 (one rounding per layer each), every contraction accumulates in f32,
// the skip sum stays in f32 registers across the layers, outputs are f32.
#pragma once
#include "conv_f16.h"

namespace mi355tts {

constexpr int WN_MAX_LAYERS = 8;
constexpr int WN_W = 64;  // columns a workgroup computes

struct WnF16Args {
  const float* h;  // [B][H][ld] f32: the block's start-conv output
  long long bs;    // floats per batch row of h / acts / skip
  int ld;
  const int* len;  // valid columns of row b: len ? len[b] : len_const
  int len_const;
  const uint4* w_in[WN_MAX_LAYERS];  // gate convs, rows paired per 32-row tile: [16 tanh rows | the same 16 channels' sigmoid rows]
  const float* b_in[WN_MAX_LAYERS];
  const uint4* w_rs[WN_MAX_LAYERS];  // res_skip 1 x 1 convs of layers 0 .. n - 2, rows in natural order [res | skip]
  const float* b_rs[WN_MAX_LAYERS];
  int n_layers;
  int margin;  // (k - 1) / 2 * n_layers: columns on either side of a tile that are recomputed, not stored
  float* acts;  // [B][H][ld]: tanh * sigmoid of the LAST layer
  float* skip;  // [B][H][ld]: sum over layers 0 .. n - 2 of their skip halves (biases included); unused when n_layers == 1
};

// NOCT = H / 8 octet rows of the hidden state; MTW = 32-row tiles a wave carries (2H / 32 tiles over 4 waves)
template <int KD, int NOCT, int MTW>
__global__ __launch_bounds__(256) void wn_f16_kernel(const WnF16Args a) {
  constexpr int H = NOCT * 8;
  constexpr int MT = NOCT / 2;    // 32-row tiles of a 2H-row conv
  constexpr int NRES = NOCT / 4;  // of which the first NRES are the res half of res_skip
  constexpr int NSL = NOCT / 2;   // 16-channel slabs of the input
  constexpr int PADC = (KD - 1) / 2;
  constexpr int HW = WN_W + 2 * PADC;  // h tile row: the computed columns + the conv's reach (zeros) on either side
  static_assert(NOCT % 4 == 0 && MTW * 4 >= MT, "tile bookkeeping");
  __shared__ uint4 hs[NOCT * HW];    // hidden state, fp16 octet rows
  __shared__ uint4 as[NOCT * WN_W];  // gated activations of the running layer

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int L = a.len ? a.len[b] : a.len_const;
  const int to = WN_W - 2 * a.margin;  // exact columns per tile
  const int t0 = blockIdx.x * to;
  if (t0 >= L) return;
  const int c_abs0 = t0 - a.margin;  // column of tile column 0
  const float* hb = a.h + (long long)b * a.bs;

  // ---- stage h: f32 rows -> fp16 octet units, zero outside the sequence; the pad columns are zeros
  for (int u = tid; u < NOCT * WN_W; u += 256) {
    const int o = u / WN_W, c = u - o * WN_W;
    const int col = c_abs0 + c;
    const bool ok = col >= 0 && col < L;
    const int cc = ok ? col : 0;
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(ok ? hb[(long long)(8 * o + e) * a.ld + cc] : 0.f);
    hs[o * HW + c + PADC] = __builtin_bit_cast(uint4, v);
  }
  for (int u = tid; u < NOCT * 2 * PADC; u += 256) {
    const int o = u / (2 * PADC), p = u - o * (2 * PADC);
    hs[o * HW + (p < PADC ? p : WN_W + p)] = uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();

  const int col = lane & 31;
  const int hi = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;
  auto aload = [&](const uint4* base, int soff) -> uint4 {  // scalar base + the lane's constant byte offset (conv_f16.h)
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + soff) + lane16);
  };
  auto bias_init = [&](floatx16 (&acc)[2], const float* bias, int mt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + mt * 32 + 8 * j + 4 * hi);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        acc[nb][4 * j + 0] = b4.x;
        acc[nb][4 * j + 1] = b4.y;
        acc[nb][4 * j + 2] = b4.z;
        acc[nb][4 * j + 3] = b4.w;
      }
    }
  };

  floatx16 acc[MTW][2];  // the running contraction of this wave's tiles x the tile's two column blocks
  floatx16 sk[MTW][2];   // skip sums of the tiles that are skip rows (f32 across the layers)
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sk[i][nb][r] = 0.f;

  for (int j = 0; j < a.n_layers; ++j) {
    const bool last = j == a.n_layers - 1;
    // ---- gate conv: 2H rows (paired), K-dim = (slab, tap); B fragments at any tap offset are one aligned ds_read_b128
    int mt[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      mt[i] = wave + 4 * i;
      bias_init(acc[i], a.b_in[j], mt[i] < MT ? mt[i] : MT - 1);
    }
    {
      const uint4* wj = a.w_in[j];
      constexpr int S = NSL * KD;
      uint4 Af[2][MTW], Bf[2][2];
      auto fetch = [&](int g, uint4 (&af)[MTW], uint4 (&bf)[2]) {
        g = g < S ? g : S - 1;
        const int s = g / KD, k = g - s * KD;
#pragma unroll
        for (int i = 0; i < MTW; ++i) af[i] = aload(wj, (((mt[i] < MT ? mt[i] : MT - 1) * NSL + s) * KD + k) * 64);
        const uint4* bp = hs + (2 * s + hi) * HW + col + k;
        bf[0] = bp[0];
        bf[1] = bp[32];
      };
      fetch(0, Af[0], Bf[0]);
      for (int g = 0; g < S; g += 2) {
        fetch(g + 1, Af[1], Bf[1]);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma_f16(Af[0][i], Bf[0][nb], acc[i][nb]);
        fetch(g + 2, Af[0], Bf[0]);
        if (g + 1 < S) {
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma_f16(Af[1][i], Bf[1][nb], acc[i][nb]);
        }
      }
    }
    // gate: rows 0 .. 15 of a tile are tanh rows, 16 .. 31 the sigmoid rows of the same channels: accumulator registers r and r + 8
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      if (mt[i] >= MT) continue;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int c = nb * 32 + col;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          float g4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ta = acc[i][nb][4 * jj + e], sb = acc[i][nb][8 + 4 * jj + e];
            const float th = 2.0f / (1.0f + expf(-2.0f * ta)) - 1.0f;
            const float sg = 1.0f / (1.0f + expf(-sb));
            g4[e] = th * sg;
          }
          if (!last) {
            half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (_Float16)g4[e];
            reinterpret_cast<uint2*>(as + (2 * mt[i] + jj) * WN_W + c)[hi] = __builtin_bit_cast(uint2, hv);
          } else {
            const int ca = c_abs0 + c;
            if (c >= a.margin && c < WN_W - a.margin && ca < L) {
              float* dst = a.acts + (long long)b * a.bs + (long long)(16 * mt[i] + 8 * jj + 4 * hi) * a.ld + ca;
#pragma unroll
              for (int e = 0; e < 4; ++e) dst[(long long)e * a.ld] = g4[e];
            }
          }
        }
      }
    }
    if (last) break;
    __syncthreads();  // the gated tile is complete; every wave has read its last h fragment
    // ---- res_skip 1 x 1: 2H rows in natural order over the gated tile
#pragma unroll
    for (int i = 0; i < MTW; ++i) bias_init(acc[i], a.b_rs[j], mt[i] < MT ? mt[i] : MT - 1);
    {
      const uint4* wj = a.w_rs[j];
      uint4 Af[2][MTW], Bf[2][2];
      auto fetch = [&](int s, uint4 (&af)[MTW], uint4 (&bf)[2]) {
        s = s < NSL ? s : NSL - 1;
#pragma unroll
        for (int i = 0; i < MTW; ++i) af[i] = aload(wj, ((mt[i] < MT ? mt[i] : MT - 1) * NSL + s) * 64);
        const uint4* bp = as + (2 * s + hi) * WN_W + col;
        bf[0] = bp[0];
        bf[1] = bp[32];
      };
      fetch(0, Af[0], Bf[0]);
      for (int s = 0; s < NSL; s += 2) {
        fetch(s + 1, Af[1], Bf[1]);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma_f16(Af[0][i], Bf[0][nb], acc[i][nb]);
        fetch(s + 2, Af[0], Bf[0]);
        if (s + 1 < NSL) {
#pragma unroll
          for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma_f16(Af[1][i], Bf[1][nb], acc[i][nb]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      if (mt[i] >= MT) continue;
      if (mt[i] < NRES) {  // res rows: h = (h + res) inside the sequence, 0 outside (layers.py:156-160 with x_mask)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int c = nb * 32 + col;
          const int ca = c_abs0 + c;
          const bool in = ca >= 0 && ca < L;
#pragma unroll
          for (int jq = 0; jq < 4; ++jq) {
            uint2* hp = reinterpret_cast<uint2*>(hs + (4 * mt[i] + jq) * HW + c + PADC) + hi;
            const half4 ho = __builtin_bit_cast(half4, *hp);
            half4 hn;
#pragma unroll
            for (int e = 0; e < 4; ++e) hn[e] = (_Float16)(in ? (float)ho[e] + acc[i][nb][4 * jq + e] : 0.f);
            *hp = __builtin_bit_cast(uint2, hn);
          }
        }
      } else {  // skip rows: summed in f32 across the layers
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sk[i][nb][r] += acc[i][nb][r];
      }
    }
    __syncthreads();  // h is updated before the next layer's gate conv reads it
  }

  if (a.n_layers > 1) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int t = wave + 4 * i;
      if (t < NRES || t >= MT) continue;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int c = nb * 32 + col;
        const int ca = c_abs0 + c;
        if (c < a.margin || c >= WN_W - a.margin || ca >= L) continue;
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          float* dst = a.skip + (long long)b * a.bs + (long long)(32 * (t - NRES) + 8 * jq + 4 * hi) * a.ld + ca;
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[(long long)e * a.ld] = sk[i][nb][4 * jq + e];
        }
      }
    }
  }
}

}  // namespace mi355tts
