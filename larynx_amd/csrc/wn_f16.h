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

// tanh(ta) * sigmoid(sb) (glow_tts/utils.py:31-38) from two v_exp_f32 and one v_rcp_f32: (1 - e^-2ta) / ((1 + e^-2ta) (1 + e^-sb)).
// ta is clamped to +-15 (tanh is 1 to 13 digits there; e^30 stays finite); a huge e^-sb gives rcp(inf) = 0 = sigmoid's limit.
__device__ __forceinline__ float gate_fast(float ta, float sb) {
  constexpr float LOG2E = 1.4426950408889634f;
  ta = fminf(fmaxf(ta, -15.0f), 15.0f);
  const float ea = __builtin_amdgcn_exp2f(-2.0f * LOG2E * ta);
  const float eb = __builtin_amdgcn_exp2f(-LOG2E * sb);
  return (1.0f - ea) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));
}

// issue order of a step (conv_f16.h's): the MTW weight loads behind the first MFMAs, then the two LDS reads, then the barrier that keeps
// the scheduler from sinking the next steps' loads to their first use (0x008 = MFMA, 0x020 = VMEM read, 0x100 = LDS read)
#define WN_STEP_ORDER()                                                          \
  do {                                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < 2 * MTW; ++q_) {                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                         \
      if (q_ < MTW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);           \
      else if (q_ - MTW < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  \
    }                                                                            \
    if (MTW < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2 - MTW, 0);        \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)

// NOCT = H / 8 octet rows of the hidden state; MTW = 32-row tiles a wave carries (2H / 32 tiles over 4 waves); AD = weight
// fragments in flight ahead of the MFMAs that use them, in steps.
//
// The weight stream.  A wave's A fragments come straight from L2 (every workgroup of the launch reads the same 3 MB of a block's
// weights; a wave owns its row tiles, so there is nothing to share through LDS), 16 bytes per lane and step and tile.  With one
// wave per SIMD nothing hides a load but the wave's own MFMAs, so the fragments ride a register ring AD steps deep, and the ring
// never drains: a layer's steps form ONE sequence — gate conv (NSL slabs x KD taps), then res_skip (NSL slabs), then the NEXT
// layer's gate conv — and step p's slot is refilled with step p + AD of that sequence, across the phase and layer boundaries (the
// layer body is fully unrolled, S + NSL steps, a multiple of AD: slot indices are compile-time).  First version: look-ahead ONE step,
// 120 us per block launch — every step waited an L2 round trip.  Biases come from LDS (staged once): a global load at a phase's
// start would sit in the in-order load counter behind the ring and drain it.
template <int KD, int NOCT, int MTW, int AD>
__global__ __launch_bounds__(256) void wn_f16_kernel(const WnF16Args a) {
  constexpr int H = NOCT * 8;
  constexpr int MT = NOCT / 2;    // 32-row tiles of a 2H-row conv
  constexpr int NRES = NOCT / 4;  // of which the first NRES are the res half of res_skip
  constexpr int NSL = NOCT / 2;   // 16-channel slabs of the input
  constexpr int PADC = (KD - 1) / 2;
  constexpr int HW = WN_W + 2 * PADC;  // h tile row: the computed columns + the conv's reach (zeros) on either side
  constexpr int S = NSL * KD;          // steps of a gate conv
  constexpr int LS = S + NSL;          // steps of a whole layer
  static_assert(NOCT % 4 == 0 && MTW * 4 >= MT, "tile bookkeeping");
  static_assert(AD <= S, "the look-ahead reaches at most into the next layer's gate conv");
  __shared__ uint4 hs[NOCT * HW];    // hidden state, fp16 octet rows
  __shared__ uint4 as[NOCT * WN_W];  // gated activations of the running layer
  __shared__ float bs[WN_MAX_LAYERS * 2 * MT * 32];  // biases: [layer][gate | res_skip][virtual row]

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
  const int n_layers = a.n_layers;

  const int col = lane & 31;
  const int hi = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;
  auto aload = [&](const uint4* base, int soff) -> uint4 {  // scalar base + the lane's constant byte offset (conv_f16.h)
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + soff) + lane16);
  };
  int mt[MTW], mtc[MTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i) {
    mt[i] = wave + 4 * i;
    mtc[i] = mt[i] < MT ? mt[i] : MT - 1;  // a wave without an i-th tile recomputes the last one and stores nothing
  }
  // the layer's step sequence: p < S gate conv step (slab p / KD, tap p % KD); p < LS res_skip slab p - S; beyond: the next layer's gate
  uint4 Af[AD + 1][MTW];
  auto afetch = [&](int p, const uint4* wg, const uint4* wr, const uint4* wn, uint4 (&af)[MTW]) {
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      if (p < S)
        af[i] = aload(wg, mtc[i] * (NSL * KD * 64) + p * 64);  // [m-tile][slab][tap][lane]: step p = slab * KD + tap
      else if (p < LS)
        af[i] = aload(wr, mtc[i] * (NSL * 64) + (p - S) * 64);
      else
        af[i] = aload(wn, mtc[i] * (NSL * KD * 64) + (p - LS) * 64);
    }
  };
  // the ring's first AD steps go out before anything else: they fly while h is staged
#pragma unroll
  for (int p = 0; p < AD; ++p) afetch(p, a.w_in[0], a.w_in[0], a.w_in[0], Af[p]);
  __builtin_amdgcn_sched_barrier(0);

  // ---- stage h: f32 rows -> fp16 octet units, zero outside the sequence; the pad columns are zeros; the biases
  static_assert((NOCT * WN_W) % 256 == 0, "whole sweeps of the 256 threads");
#pragma unroll 2
  for (int u = tid; u < NOCT * WN_W; u += 256) {  // clamped addresses, nothing behind a branch: a unit's eight loads fly together
    const int o = u / WN_W, c = u - o * WN_W;
    const int cl = c_abs0 + c;
    const bool ok = cl >= 0 && cl < L;
    const float* src = hb + (long long)(8 * o) * a.ld + (ok ? cl : 0);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = src[(long long)e * a.ld];
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (_Float16)(ok ? f[e] : 0.f);
    hs[o * HW + c + PADC] = __builtin_bit_cast(uint4, v);
  }
  for (int u = tid; u < NOCT * 2 * PADC; u += 256) {
    const int o = u / (2 * PADC), p = u - o * (2 * PADC);
    hs[o * HW + (p < PADC ? p : WN_W + p)] = uint4{0u, 0u, 0u, 0u};
  }
  for (int u = tid; u < n_layers * 2 * MT * 32; u += 256) {
    const int j = u / (2 * MT * 32), r = u - j * (2 * MT * 32);
    const bool rs = r >= MT * 32;
    const float* src = rs ? (j < n_layers - 1 ? a.b_rs[j] + (r - MT * 32) : a.b_in[j]) : a.b_in[j] + r;
    bs[u] = *src;  // (the last layer has no res_skip here: its slot is never read)
  }
  __syncthreads();

  auto bias_init = [&](floatx16 (&acc)[2], const float* bias) {  // bias: the tile's 32 rows in LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + 8 * j + 4 * hi);
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
  // skip sums of the tiles that are skip rows (f32 across the layers).  Tile i of wave w is row tile w + 4 i: the tiles i < SK0 are
  // res rows on every wave and carry no sum
  constexpr int SK0 = NRES / 4;
  floatx16 sk[MTW - SK0][2];
#pragma unroll
  for (int i = 0; i < MTW - SK0; ++i)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) sk[i][nb][r] = 0.f;

  for (int j = 0; j < n_layers; ++j) {
    const bool last = j == n_layers - 1;
    // the sequence's pointers; past the last layer's gate conv the ring is refilled from addresses nobody uses (no load behind a branch)
    const uint4* wg = a.w_in[j];
    const uint4* wr = last ? wg : a.w_rs[j];
    const uint4* wn = last ? wg : a.w_in[j + 1];
    // ---- gate conv: 2H rows (paired), K-dim = (slab, tap); B fragments at any tap offset are one aligned ds_read_b128
#pragma unroll
    for (int i = 0; i < MTW; ++i) bias_init(acc[i], bs + (j * 2 * MT + mtc[i]) * 32);
    {
      uint4 Bf[2][2];
      auto bfetch = [&](int g, uint4 (&bf)[2]) {
        const int s = g / KD, k = g - s * KD;
        const uint4* bp = hs + (2 * s + hi) * HW + col + k;
        bf[0] = bp[0];
        bf[1] = bp[32];
      };
      bfetch(0, Bf[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < S; ++g) {
        afetch(g + AD, wg, wr, wn, Af[AD]);
        bfetch(g + 1 < S ? g + 1 : g, Bf[1]);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma_f16(Af[0][i], Bf[0][nb], acc[i][nb]);
        WN_STEP_ORDER();
#pragma unroll
        for (int d = 0; d < AD; ++d)
#pragma unroll
          for (int i = 0; i < MTW; ++i) Af[d][i] = Af[d + 1][i];
        Bf[0][0] = Bf[1][0];
        Bf[0][1] = Bf[1][1];
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
          for (int e = 0; e < 4; ++e) g4[e] = gate_fast(acc[i][nb][4 * jj + e], acc[i][nb][8 + 4 * jj + e]);
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
    for (int i = 0; i < MTW; ++i) bias_init(acc[i], bs + ((j * 2 + 1) * MT + mtc[i]) * 32);
    {
      uint4 Bf[2][2];
      auto bfetch = [&](int s, uint4 (&bf)[2]) {
        const uint4* bp = as + (2 * s + hi) * WN_W + col;
        bf[0] = bp[0];
        bf[1] = bp[32];
      };
      bfetch(0, Bf[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
        afetch(S + s + AD, wg, wr, wn, Af[AD]);
        bfetch(s + 1 < NSL ? s + 1 : s, Bf[1]);
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma_f16(Af[0][i], Bf[0][nb], acc[i][nb]);
        WN_STEP_ORDER();
#pragma unroll
        for (int d = 0; d < AD; ++d)
#pragma unroll
          for (int i = 0; i < MTW; ++i) Af[d][i] = Af[d + 1][i];
        Bf[0][0] = Bf[1][0];
        Bf[0][1] = Bf[1][1];
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
      } else if (i >= SK0) {  // skip rows: summed in f32 across the layers
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sk[i - SK0][nb][r] += acc[i][nb][r];
      }
    }
    __syncthreads();  // h is updated before the next layer's gate conv reads it
  }

  if (n_layers > 1) {
#pragma unroll
    for (int i = SK0; i < MTW; ++i) {
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
          for (int e = 0; e < 4; ++e) dst[(long long)e * a.ld] = sk[i - SK0][nb][4 * jq + e];
        }
      }
    }
  }
}

}  // namespace mi355tts
