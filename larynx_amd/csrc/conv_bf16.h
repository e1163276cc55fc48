// conv1d as an implicit GEMM on the CDNA4 bf16 matrix cores with SPLIT operands — the
// reduced-precision mode behind the reference's `half` switch (larynx/glow_tts.py:90-91,
// larynx/hifi_gan.py:96-97, README "--half"), SURVEY.md §8(f) rank 4.
//
// Every f32 operand x is split on the fly into two bf16 numbers, x = hi + lo + O(2^-17 |x|)
// (hi = bf16(x), lo = bf16(x - hi)), and a product is formed from three bf16 MFMAs accumulated in
// f32:   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (the dropped terms are O(2^-16 |a b|)).
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32 MFMA the exact path uses
// (MI355X_MICROARCH.md: ~2.5 PFLOP/s dense vs 157 TFLOP/s), so three of them still leave 5.3x the
// matrix rate — the kernel is bound by operand delivery (LDS reads, the weight stream), not by
// the matrix pipe.  Accuracy: ~1e-5 relative per layer, i.e. far better than the fp16 tensors
// of the reference's `--half` (11-bit mantissa everywhere, no f32 accumulate guarantee).
//
//   y[co][t] = [y +] alpha * (bias[co] + res[co][t] + sum_{ci,k} W[co][ci][k] * lrelu(x[ci][t + k*dil - pad]))
//
// GEMM view as in conv_mfma.h: M = output channels (A = weights, pre-split and pre-packed in fragment
// order, streamed L2 -> VGPR 16 B per lane), N = time (B = activations, split once per element while
// they are staged into LDS in an [octet][column][8 channels] bf16 layout, so one B fragment — 8
// consecutive channels of one column — is ONE aligned ds_read_b128), K-dim = (16-channel slab, tap).
// One workgroup = WM x WN waves; a wave owns MB x NB blocks of 32 x 32 over ALL input channels
// (no k-split: 16-deep MFMAs make the per-wave work large enough).  Used for the HiFi-GAN ResBlock
// convs (hifi_gan/models.py:91-98, 136-141) when the vocoder was switched to this mode.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_mfma.h"

namespace mi355tts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two floats -> two bf16 (round to nearest even), first argument in the low half
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  bf16x2 p = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, p);
}
// hi/lo split of two floats: x = hi + lo + O(2^-17 |x|)
__device__ __forceinline__ void split_bf16(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16(a, b);
  const float ah = __uint_as_float(hi << 16), bh = __uint_as_float(hi & 0xffff0000u);
  lo = pack_bf16(a - ah, b - bh);
}
__device__ __forceinline__ floatx16 mfma_bf16(const uint4& a, const uint4& b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// LDS of one workgroup, in uint4 (16-byte) units: two buffers x {hi, lo} x 4 octets x XW columns
template <int NB, int WN, int HALO>
constexpr int conv_bf16_lds_units() {
  return 2 * 2 * 4 * (32 * NB * WN + HALO);
}

// KS = 2 splits the K-dim between two groups of waves — group kg takes slab kg of every 32-channel chunk — and sums
// the two partial tiles through LDS: launches with too few tiles to fill the chip (stage 0 of 'high' at batch 1:
// 256 channels x 4992 columns = 78 tiles of 128 x 128 per conv) then put 8 waves on every tile instead of 4.
// EPI = EPI_LINEAR (ResBlock convs) or EPI_UPSAMPLE (the polyphase ConvTranspose1d of conv_mfma.h: virtual rows co * u + r
// scattered to y[co][q * u + r - up_pad]; its input may be the MRF average (x + x2 + x3) / in_div of the previous stage).
template <int K, int MB, int NB, int WM, int WN, int HALO, int TERMS, int KS = 1, int EPI = EPI_LINEAR>
__device__ __forceinline__ void conv_bf16_tile(const ConvArgs& a, const int tile_x, const int tile_y, const int b, uint4* __restrict__ xs) {
  static_assert(KS == 1 || KS == 2, "k-split of the bf16 tile is 1 or 2");
  static_assert(EPI == EPI_LINEAR || EPI == EPI_UPSAMPLE, "bf16 tile epilogues");
  constexpr int NWAVES = WM * WN * KS;
  constexpr int NT = 64 * NWAVES;
  constexpr int T_T = 32 * NB * WN;   // time columns per workgroup
  constexpr int XW = T_T + HALO;      // staged columns per octet row
  constexpr int XQ = XW / 4;          // column quads
  constexpr int OCT = 4;              // 32 input channels per staged chunk = 2 MFMA slabs of 16
  constexpr int PLANE = OCT * XW;     // uint4 units per plane (hi or lo)
  constexpr int BUF = 2 * PLANE;      // per buffer
  constexpr int NUNITS = OCT * XQ;    // staging units (octet, column quad) per chunk
  constexpr int NU = (NUNITS + NT - 1) / NT;
  static_assert(XW % 4 == 0 && (TERMS == 1 || TERMS == 3), "bad tile parameters");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = (wave / WM) % WN;
  const int kg = wave / (WM * WN);
  const int t0 = tile_x * T_T;
  const int mt0 = (tile_y * WM + wm) * MB;

  const int Lin = a.in_len ? a.in_len[b] * a.in_mul : a.in_const;
  const int Lout = a.out_len ? a.out_len[b] * a.out_mul : a.out_const;
  const int n_len = (EPI == EPI_UPSAMPLE) ? (Lin > 0 ? Lin + (K - 1) : 0) : Lout;  // extent of the GEMM's N axis
  if (t0 >= n_len) return;  // uniform per workgroup

  const float slope = a.in_slope;
  const float* xb = a.x + (long long)b * a.x_bs;
  const float* xb2 = a.x2 ? a.x2 + (long long)b * a.x_bs : nullptr;
  const float* xb3 = a.x3 ? a.x3 + (long long)b * a.x_bs : nullptr;
  const int nchunks = (a.Cin + 31) / 32;
  const int PA = (a.pad + 3) & ~3;
  const int cin_last = a.Cin - 1;
  const int ld_last4 = a.x_ld - 4;

  // ---- staging: unit (octet o, column quad q) = 8 channels x 4 columns, eight 16-byte loads (one per
  // channel, coalesced along time), split into bf16 hi/lo and stored column by column as 16-byte
  // [8 channels] groups.  The loads of chunk c+1 are issued before chunk c's MFMA phase and consumed after it.
  auto gload = [&](int chunk, float4 (&pre)[NU][8]) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = tid + NT * i;
      const int o = u < NUNITS ? u / XQ : 0, q = u < NUNITS ? u - (u / XQ) * XQ : 0;
      const int c0 = t0 - PA + 4 * q;
      const int cc = c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = chunk * 32 + 8 * o + j;
        pre[i][j] = *reinterpret_cast<const float4*>(xb + (long long)(ci < a.Cin ? ci : cin_last) * a.x_ld + cc);
      }
      if (xb2) {  // wave-uniform: the MRF average of the previous stage's chain outputs, taken on load (this path waits)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ci = chunk * 32 + 8 * o + j;
          const long long off = (long long)(ci < a.Cin ? ci : cin_last) * a.x_ld + cc;
          float4 t2 = *reinterpret_cast<const float4*>(xb2 + off);
          float4& p = pre[i][j];
          p.x += t2.x; p.y += t2.y; p.z += t2.z; p.w += t2.w;
          if (xb3) {
            t2 = *reinterpret_cast<const float4*>(xb3 + off);
            p.x += t2.x; p.y += t2.y; p.z += t2.z; p.w += t2.w;
          }
          p.x = p.x / a.in_div; p.y = p.y / a.in_div; p.z = p.z / a.in_div; p.w = p.w / a.in_div;
        }
      }
    }
  };
  // leaky ReLU as max(v, v * slope) when 0 <= slope <= 1 (two VALU ops instead of three; same bits for finite v)
  const bool slope01 = slope >= 0.f && slope <= 1.f;
  // a tile whose staged window lies inside the row needs no column masks (the common case by far)
  const bool cols_inside = t0 - PA >= 0 && t0 - PA + XW <= Lin;
  auto lstore = [&](int buf, int chunk, const float4 (&pre)[NU][8]) {
    const bool inside = cols_inside && slope01 && chunk * 32 + 32 <= a.Cin;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = tid + NT * i;
      if (u >= NUNITS) continue;
      const int o = u / XQ, q = u - o * XQ;
      const int c0 = t0 - PA + 4 * q;
      float v[8][4];
      if (inside) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 p = pre[i][j];
          v[j][0] = fmaxf(p.x, p.x * slope);
          v[j][1] = fmaxf(p.y, p.y * slope);
          v[j][2] = fmaxf(p.z, p.z * slope);
          v[j][3] = fmaxf(p.w, p.w * slope);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool cok = chunk * 32 + 8 * o + j < a.Cin;
          const float4 p = pre[i][j];
          v[j][0] = (cok && c0 >= 0 && c0 < Lin) ? p.x : 0.f;
          v[j][1] = (cok && c0 + 1 >= 0 && c0 + 1 < Lin) ? p.y : 0.f;
          v[j][2] = (cok && c0 + 2 >= 0 && c0 + 2 < Lin) ? p.z : 0.f;
          v[j][3] = (cok && c0 + 3 >= 0 && c0 + 3 < Lin) ? p.w : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] > 0.f ? v[j][e] : v[j][e] * slope;
        }
      }
      uint4* dst = xs + buf * BUF + o * XW + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint4 hi, lo;
        split_bf16(v[0][e], v[1][e], hi.x, lo.x);
        split_bf16(v[2][e], v[3][e], hi.y, lo.y);
        split_bf16(v[4][e], v[5][e], hi.z, lo.z);
        split_bf16(v[6][e], v[7][e], hi.w, lo.w);
        dst[e] = hi;
        dst[PLANE + e] = lo;
      }
    }
  };

  floatx16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  float4 pre[NU][8];

  // A stream of m-tile mt: uint4 index (((mt*nslab + slab)*K + k)*2 + plane)*64 + lane
  const uint4* wq[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) wq[mb] = reinterpret_cast<const uint4*>(a.w16) + (long long)(mt0 + mb) * a.nslab * K * 128 + lane;
  constexpr int SPC = 2 / KS;  // slabs of a chunk this k-group walks
  constexpr int S = SPC * K;   // steps per chunk: tap-major, SPC slabs per tap
  const int last_step = nchunks * S - 1;
  auto a_off = [&](int g) -> int {  // uint4 offset of global step g (clamped at the end: harmless re-load)
    g = g < last_step ? g : last_step;
    const int ch = g / S, st = g - ch * S;
    const int k = st / SPC, s = KS == 2 ? kg : (st & 1);
    return ((ch * 2 + s) * K + k) * 128;
  };
  // Operand pipeline of one wave (registers): weight fragments run AD steps ahead of the MFMAs that use them
  // (Ah[0] = this step ... Ah[AD] = in flight for step + AD: an L2 hit costs more than one step of 12 MFMAs —
  // measured -7 % going from one step ahead to two), LDS fragments ONE step ahead (B0 = this step, B1 = next).  The
  // fillers are pinned behind individual MFMAs with sched_group_barrier — left alone the compiler sinks every load
  // to just above its first use and the wave stalls once per step on LDS and once on L2 (-20 % on the class).
#ifndef BF16_ADIST
#define BF16_ADIST 2
#endif
  constexpr int AD = BF16_ADIST;
  uint4 Ah[AD + 1][MB], Al[AD + 1][MB];
  uint4 B0h[NB], B0l[NB], B1h[NB], B1l[NB];

  gload(0, pre);
#pragma unroll
  for (int d = 0; d < AD; ++d)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      Ah[d][mb] = wq[mb][a_off(d)];
      if (TERMS == 3) Al[d][mb] = wq[mb][a_off(d) + 64];
    }
  lstore(0, 0, pre);
  __syncthreads();

  // this lane's B column inside the staged row, and its octet half
  const int colb = wn * (NB * 32) + (lane & 31) + (PA - a.pad);
  const int ohalf = lane >> 5;
  constexpr int NMF = TERMS * MB * NB;  // MFMAs per step

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    const bool more = chunk + 1 < nchunks;
    if (more) gload(chunk + 1, pre);
    const uint4* xt = xs + buf * BUF + colb;
    auto bread = [&](int st, uint4* bh, uint4* bl) {
      const int k = st / SPC, s = KS == 2 ? kg : (st & 1);
      const uint4* bp = xt + (2 * s + ohalf) * XW + k * a.dil;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bh[nb] = bp[nb * 32];
        if (TERMS == 3) bl[nb] = bp[PLANE + nb * 32];
      }
    };
    bread(0, B0h, B0l);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < S; ++st) {
      {
        const int off = a_off(chunk * S + st + AD);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          Ah[AD][mb] = wq[mb][off];
          if (TERMS == 3) Al[AD][mb] = wq[mb][off + 64];
        }
      }
      if (st + 1 < S) bread(st + 1, B1h, B1l);
      // term-major order: consecutive MFMAs go to different accumulators
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma_bf16(Ah[0][mb], B0h[nb], acc[mb][nb]);
      if (TERMS == 3) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma_bf16(Ah[0][mb], B0l[nb], acc[mb][nb]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma_bf16(Al[0][mb], B0h[nb], acc[mb][nb]);
      }
      // issue order: one filler behind each MFMA (0x008 = MFMA, 0x020 = VMEM read, 0x100 = LDS read)
      {
        constexpr int NV = (TERMS == 3 ? 2 : 1) * MB, ND = (TERMS == 3 ? 2 : 1) * NB;
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          else if (st + 1 < S) {
            // one LDS read behind each remaining MFMA, whatever is left over behind the last one
            constexpr int R = NMF - NV;
            const int j = i - NV;
            const int cnt = j < R - 1 ? (j < ND ? 1 : 0) : (ND - (R - 1) > 0 ? ND - (R - 1) : (j < ND ? 1 : 0));
            if (cnt == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (cnt == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else if (cnt == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            else if (cnt == 4) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int d = 0; d < AD; ++d)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          Ah[d][mb] = Ah[d + 1][mb];
          if (TERMS == 3) Al[d][mb] = Al[d + 1][mb];
        }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        B0h[nb] = B1h[nb];
        if (TERMS == 3) B0l[nb] = B1l[nb];
      }
    }
    if (more) lstore(buf ^ 1, chunk + 1, pre);
    __syncthreads();
  }

  if constexpr (KS == 2) {
    // sum the two k-groups' partial tiles through LDS (the staging buffers are free: the loop ended on a barrier),
    // one column block per round; group 0 owns the epilogue
    float* red = reinterpret_cast<float*>(xs);
    static_assert(WM * WN * MB * 16 * 64 * 4 <= conv_bf16_lds_units<NB, WN, HALO>() * 16, "reduction scratch must fit the staging buffers");
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nb > 0) __syncthreads();
      if (kg == 1) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((wn * WM + wm) * MB + mb) * 16 + r) * 64 + lane] = acc[mb][nb][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][nb][r] += red[(((wn * WM + wm) * MB + mb) * 16 + r) * 64 + lane];
      }
    }
    if (kg != 0) return;
  }

  const int col = lane & 31;
  const int rbase = 4 * (lane >> 5);
  if constexpr (EPI == EPI_UPSAMPLE) {
    // polyphase scatter (conv_mfma.h): registers 4g .. 4g+3 of a lane are 4 consecutive virtual rows = 4 consecutive phases
    // of ONE output channel = 4 consecutive output samples: one float4 store where the alignment allows
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float bb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (mt0 + mb) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        bb[r] = (a.bias && row < a.rows) ? a.bias[row] : 0.f;
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int q = t0 + (wn * NB + nb) * 32 + col;
        if (q >= n_len) continue;
        const bool vec = (a.up & 3) == 0 && (a.up_pad & 3) == 0 && (a.y_ld & 3) == 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int row0 = (mt0 + mb) * 32 + 8 * g4 + rbase;
          if (row0 >= a.rows) continue;
          if (vec) {
            const int co = row0 / a.up;
            const int n0 = q * a.up + (row0 - co * a.up) - a.up_pad;
            float* dst = a.y + (long long)b * a.y_bs + (long long)co * a.y_ld + n0;
            if (n0 >= 0 && n0 + 3 < Lout) {
              *reinterpret_cast<float4*>(dst) = make_float4(acc[mb][nb][4 * g4] + bb[4 * g4], acc[mb][nb][4 * g4 + 1] + bb[4 * g4 + 1],
                                                            acc[mb][nb][4 * g4 + 2] + bb[4 * g4 + 2], acc[mb][nb][4 * g4 + 3] + bb[4 * g4 + 3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n0 + e >= 0 && n0 + e < Lout) dst[e] = acc[mb][nb][4 * g4 + e] + bb[4 * g4 + e];
            }
          } else if (a.up == 2) {
            // stride 2: registers (e, e + 1), e even, are the two phases of one channel = two consecutive samples: one
            // 8-byte (4-byte aligned) store per pair, as in conv_mfma.h
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const int row = row0 + e;  // even
              if (row >= a.rows) continue;
              const int n0 = q * 2 - a.up_pad;
              float* dst = a.y + (long long)b * a.y_bs + (long long)(row >> 1) * a.y_ld + n0;
              const float v0 = acc[mb][nb][4 * g4 + e] + bb[4 * g4 + e], v1 = acc[mb][nb][4 * g4 + e + 1] + bb[4 * g4 + e + 1];
              if (n0 >= 0 && n0 + 1 < Lout && row + 1 < a.rows) {
                typedef float up_float2 __attribute__((ext_vector_type(2), aligned(4)));
                *reinterpret_cast<up_float2*>(dst) = up_float2{v0, v1};
              } else {
                if (n0 >= 0 && n0 < Lout) dst[0] = v0;
                if (row + 1 < a.rows && n0 + 1 >= 0 && n0 + 1 < Lout) dst[1] = v1;
              }
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = row0 + e;
              if (row >= a.rows) continue;
              const int co = row / a.up;
              const int n = q * a.up + (row - co * a.up) - a.up_pad;
              if (n >= 0 && n < Lout) a.y[(long long)b * a.y_bs + (long long)co * a.y_ld + n] = acc[mb][nb][4 * g4 + e] + bb[4 * g4 + e];
            }
          }
        }
      }
    }
    return;
  }
  // ---- epilogue: bias, residual, scale, accumulate; loads batched from clamped addresses
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float bb[16];
    int roff[16];
    bool rok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (mt0 + mb) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
      rok[r] = row < a.rows;
      bb[r] = (a.bias && rok[r]) ? a.bias[row] : 0.f;
      roff[r] = (rok[r] ? row : a.rows - 1) * a.y_ld;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int t = t0 + (wn * NB + nb) * 32 + col;
      const bool tok = t < Lout;
      const int tc = tok ? t : Lout - 1;
      float* yb = a.y + (long long)b * a.y_bs + tc;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[mb][nb][r] + bb[r];
      if (a.res) {
        const float* rb = a.res + (long long)b * a.y_bs + tc;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = rb[roff[r]];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += rv[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] *= a.alpha;
      if (a.accum) {
        float ov[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ov[r] = yb[roff[r]];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += ov[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (rok[r] && tok) yb[roff[r]] = v[r];
    }
  }
}

template <int K, int MB, int NB, int WM, int WN, int HALO, int TERMS, int KS = 1, int EPI = EPI_LINEAR>
__global__ __launch_bounds__(64 * WM * WN * KS) void conv_bf16_kernel(const ConvArgs a) {
  __shared__ uint4 xs[conv_bf16_lds_units<NB, WN, HALO>()];
  int tile_x, tile_y;
  int gx = gridDim.x;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (conv_mfma.h, row_tiles)
    gx = row_tiles(conv_n_len<K, EPI>(a, blockIdx.z), 32 * NB * WN);
    if (lin >= gx * (int)gridDim.y) return;
  }
  xcd_tile_lin(lin, gx, gridDim.y, tile_x, tile_y, a.rows_major);
  conv_bf16_tile<K, MB, NB, WM, WN, HALO, TERMS, KS, EPI>(a, tile_x, tile_y, blockIdx.z, xs);
}

// The three MRF chains' same-geometry convs in ONE launch (see conv_group_kernel).
template <int K0, int K1, int K2, int MB, int NB, int WM, int WN, int H0, int H1, int H2, int TERMS, int KS = 1>
#ifndef BF16_OCC
#define BF16_OCC 1
#endif
__global__ __launch_bounds__(64 * WM * WN * KS, (NB == 4 && KS == 1) ? BF16_OCC : 1) void conv_bf16_group_kernel(const ConvGroupArgs g) {
  constexpr int L0 = conv_bf16_lds_units<NB, WN, H0>(), L1 = conv_bf16_lds_units<NB, WN, H1>(), L2 = conv_bf16_lds_units<NB, WN, H2>();
  __shared__ uint4 xs[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  const bool ragged = gridDim.z > 1;  // a row deals only its own tiles (conv_mfma.h, row_tiles)
  constexpr int T_T = 32 * NB * WN;
  int tx, ty;
  if (lin < g.off[1]) {
    const int gx = ragged ? row_tiles(conv_n_len<K0, EPI_LINEAR>(g.c[0], b), T_T) : g.gx[0];
    if (lin >= gx * g.gy[0]) return;
    xcd_tile_lin(lin, gx, g.gy[0], tx, ty);
    conv_bf16_tile<K0, MB, NB, WM, WN, H0, TERMS, KS>(g.c[0], tx, ty, b, xs);
  } else if (lin < g.off[2]) {
    const int l = lin - g.off[1];
    const int gx = ragged ? row_tiles(conv_n_len<K1, EPI_LINEAR>(g.c[1], b), T_T) : g.gx[1];
    if (l >= gx * g.gy[1]) return;
    xcd_tile_lin(l, gx, g.gy[1], tx, ty);
    conv_bf16_tile<K1, MB, NB, WM, WN, H1, TERMS, KS>(g.c[1], tx, ty, b, xs);
  } else {
    const int l = lin - g.off[2];
    const int gx = ragged ? row_tiles(conv_n_len<K2, EPI_LINEAR>(g.c[2], b), T_T) : g.gx[2];
    if (l >= gx * g.gy[2]) return;
    xcd_tile_lin(l, gx, g.gy[2], tx, ty);
    conv_bf16_tile<K2, MB, NB, WM, WN, H2, TERMS, KS>(g.c[2], tx, ty, b, xs);
  }
}

}  // namespace mi355tts
