// The ResBlock convs of the wide HiFi-GAN stages (hifi_gan/models.py:91-98: conv1 of a dilation step, k = 3 / 7 / 11,
// d = 1 / 3 / 5, and conv2, d = 1, + the residual) on the 128-row tile with a CONTINUOUS matrix stream.
//
// conv_tile (conv_mfma.h) walks the K-depth in staged chunks and ends every chunk on "activation tile -> LDS, wait,
// barrier, first operand reads": measured on the device (tools/probe/rb_diag.hip, profiles/NOTES.md round 4), a wave
// of the 128-row tile spends 5-6 k cycles per chunk outside its MFMA loop — 1.75x the 3 k cycles of MFMA issue a
// k = 3 chunk holds — and four workgroups per CU do not hide that: the k = 3 member ran its chunks at 0.77 of the
// matrix pipe.  Here the chunk seam is gone:
//   * the LDS ring has FOUR buffers; the activation tile of chunk c + 2 is written during the first steps of chunk c
//     (it was requested from memory during chunk c - 1), so nothing of the staging sits between two chunks' MFMAs;
//   * ONE barrier per chunk, placed after those stores in the middle of the chunk: it publishes chunk c + 2 a whole
//     chunk before its first read, and a wave that reaches it has finished chunk c - 1, so the buffer the stores of
//     chunk c + 1 go to ((c + 3) & 3 = (c - 1) & 3) is free — every wave has about a chunk of slack on either side,
//     the barrier hardly ever waits;
//   * the operand reads of the next step run across the chunk boundary (the next buffer was published a chunk ago).
// Same tile, same fragment streams, same accumulation order as conv_tile<K, 16, 1, 2, 1, 1, HALO, EPI_LINEAR, 4>:
// results are bit-identical to the chunked kernel (tests compare them).
#pragma once
#include "conv_mfma.h"

namespace mi355tts {

#ifndef RB_STAMP
#define RB_STAMP(n)
#endif
#ifndef RB_PRIO  // wave priority outside the main loop (prologue / epilogue of a tile that starts among running ones)
#define RB_PRIO 3
#endif
#ifndef RB_WIDE_STORES  // interior tiles store 16 bytes per lane after a transpose through LDS (0 = A/B builds: dword stores)
#define RB_WIDE_STORES 1
#endif
#ifndef RB_ABL  // probe builds only (tools/probe/rb_diag.hip; results are WRONG when set): ablation bit mask
#define RB_ABL 0
#endif

// ring buffer stride: whole sweeps of the 256 threads (a thread past the tile's last float4 stores into padding)
template <int HALO, int NB = 2, int CI_C = 16>
constexpr int rb_buf_floats() { return ((CI_C * (32 * NB + HALO) / 4 + 255) / 256) * 256 * 4; }
template <int HALO, int NB = 2, int CI_C = 16>
constexpr int rb_lds_floats() { return 4 * rb_buf_floats<HALO, NB, CI_C>(); }

// One workgroup (4 waves = 4 row groups of 32 rows) = rows [128 tile_y, +128) x columns [64 tile_x, +64) of batch row b.
// EPI_LINEAR: ConvArgs x (no x2 / x3), bias, res, y; alpha = 1, no accumulate / activation / row split (the host checks).
// EPI_UPSAMPLE: the polyphase ConvTranspose1d of conv_tile (K = 2 taps, rows = C_out * u virtual rows scattered to
// q u + r - pad), optionally with MRF = the consumer-side MRF average: input = ((x + x2) + x3) / in_div, summed when the
// tile is written to LDS — the three planes' loads go out together a chunk earlier, nothing waits for them.
// CI_C = channels per staged chunk (16 everywhere).  32 was measured for the two-tap upsamplers, whose 16-channel chunk is only
// 32 MFMAs per wave between two chunk seams: 48 KB of LDS and 141 VGPRs instead of 32 KB / 101 for -2 us per utterance
// (profiles/NOTES.md, round 5) — not instantiated.  The MFMA sequence does not depend on it: same bits.
template <int K, int HALO, int EPI = EPI_LINEAR, bool MRF = false, int NB = 2, int CI_C = 16>
__device__ __forceinline__ void rb_tile(const ConvArgs& a, const int tile_x, const int tile_y, const int b, float* __restrict__ xs) {
  static_assert(EPI == EPI_LINEAR || EPI == EPI_UPSAMPLE, "rb_tile: linear and upsample epilogues");
  constexpr int NT = 256, T_T = 32 * NB;  // NB column blocks of 32 per wave: 64-column tiles, or 32 (NB = 1) where a
  // launch would otherwise have too few workgroups (the 256-channel stage at batch 1)
  constexpr int XW = T_T + HALO, XW4 = XW / 4, OCTS = CI_C / 8, S = OCTS * K;
  constexpr int NF4 = CI_C * XW4, NE = (NF4 + NT - 1) / NT;
  constexpr int BUF = rb_buf_floats<HALO, NB, CI_C>();  // floats per ring buffer (>= CI_C * XW)
  static_assert(BUF == NE * NT * 4, "ring stride = whole thread sweeps");
  static_assert(XW % 4 == 0 && S >= NE + 2, "bad tile parameters");

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int t0 = tile_x * T_T;
  const int mt0 = tile_y * 4 + wm;
  const int Lin = a.in_len ? a.in_len[b] * a.in_mul : a.in_const;
  const int Lout = a.out_len ? a.out_len[b] * a.out_mul : a.out_const;
  const int n_len = (EPI == EPI_UPSAMPLE) ? (Lin > 0 ? Lin + (K - 1) : 0) : Lout;  // extent of the GEMM's N axis
  if (t0 >= n_len) return;  // uniform per workgroup

  const float slope = a.in_slope;
  const float* xb = a.x + (long long)b * a.x_bs;
  const int nchunks = ((a.Cin + 7) / 8 + OCTS - 1) / OCTS;
  const int PA = (a.pad + 3) & ~3;
  const int used4 = (PA - a.pad + T_T + (K - 1) * a.dil + 3) >> 2;
  const int cin_last = a.Cin - 1, ld_last4 = a.x_ld - 4;

  // staging: as conv_tile (16-byte loads from clamped addresses, mask + leaky-ReLU on the way into LDS), one register set
  constexpr int NP = MRF ? 3 : 1;  // input planes
  const float* xb2 = MRF ? a.x2 + (long long)b * a.x_bs : nullptr;
  const float* xb3 = MRF ? a.x3 + (long long)b * a.x_bs : nullptr;
  float4 pre[NE * NP];
  auto gload = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / XW4, f = e - row * XW4;
      const int ci = chunk * CI_C + (row < CI_C ? row : CI_C - 1);
      const int c0 = t0 - PA + 4 * f;
      const int off = (ci < a.Cin ? ci : cin_last) * a.x_ld + (c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0));
      pre[i] = *reinterpret_cast<const float4*>(xb + off);
      if constexpr (MRF) {
        pre[NE + i] = *reinterpret_cast<const float4*>(xb2 + off);
        pre[2 * NE + i] = *reinterpret_cast<const float4*>(xb3 + off);
      }
    }
  };
  auto lstore1 = [&](int buf, int chunk, int i) {
    const int e = tid + NT * i;
    const int row = e / XW4, f = e - row * XW4;
    const int c0 = t0 - PA + 4 * f;
    const bool fok = chunk * CI_C + row < a.Cin && f < used4;
    float4 v = pre[i];
    if constexpr (MRF) {  // xs / num_kernels of hifi_gan/models.py:191-197, in conv_tile's order: ((x + x2) + x3) / in_div
      const float4 v2 = pre[NE + i], v3 = pre[2 * NE + i];
      v.x = ((v.x + v2.x) + v3.x) / a.in_div;
      v.y = ((v.y + v2.y) + v3.y) / a.in_div;
      v.z = ((v.z + v2.z) + v3.z) / a.in_div;
      v.w = ((v.w + v2.w) + v3.w) / a.in_div;
    }
    v.x = (fok && c0 >= 0 && c0 < Lin) ? v.x : 0.f;
    v.y = (fok && c0 + 1 >= 0 && c0 + 1 < Lin) ? v.y : 0.f;
    v.z = (fok && c0 + 2 >= 0 && c0 + 2 < Lin) ? v.z : 0.f;
    v.w = (fok && c0 + 3 >= 0 && c0 + 3 < Lin) ? v.w : 0.f;
    v.x = v.x > 0.f ? v.x : v.x * slope;
    v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope;
    v.w = v.w > 0.f ? v.w : v.w * slope;
    reinterpret_cast<float4*>(xs + buf * BUF)[e] = v;  // e >= NF4: padding of the ring buffer
  };

  floatx16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  // A fragments: float4 index ((mt * noct + oct) * K + k) * 64 + lane, step q of chunk c = octet 2 c + q / K, tap q % K
  const float4* wq = reinterpret_cast<const float4*>(a.w) + (long long)mt0 * a.noct * K * 64 + lane;
  const int last_step = nchunks * S - 1;
  auto a_index = [&](int chunk, int q) -> long long {
    int g = chunk * S + q;
    g = g < last_step ? g : last_step;
    return (long long)g * 64;  // (octet, tap) pairs are consecutive: g = (chunk * OCTS + oi) * K + k
  };
  constexpr int RD = MI355TTS_ARING;
  float4 ar[RD];

  // prologue: chunks 0 and 1 staged, chunk 2 requested, the first fragments requested — one batch of loads.
  // The issue arbiter serves waves by priority, then age: a tile that starts while the CU's other workgroups are in
  // their MFMA loops is the youngest everywhere and gets the leftover issue slots (measured: 18-40 k cycles for a
  // prologue that takes 4 k on an idle CU), so the prologue and the epilogue run at raised priority.
  RB_STAMP(0);
  if (RB_PRIO) __builtin_amdgcn_s_setprio(RB_PRIO);
  gload(0);
#pragma unroll
  for (int i = 0; i < RD - 1; ++i) ar[i] = wq[a_index(0, i)];
  {
    constexpr int NR = NE * NP;
    float4 p0[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) p0[i] = pre[i];
    gload(1);
    float4 p1[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) p1[i] = pre[i];
    gload(2);
    float4 p2[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { p2[i] = pre[i]; pre[i] = p0[i]; }
#pragma unroll
    for (int i = 0; i < NE; ++i) lstore1(0, 0, i);
#pragma unroll
    for (int i = 0; i < NR; ++i) pre[i] = p1[i];
#pragma unroll
    for (int i = 0; i < NE; ++i) lstore1(1, 1, i);
#pragma unroll
    for (int i = 0; i < NR; ++i) pre[i] = p2[i];
  }
  __syncthreads();
  if (RB_PRIO) __builtin_amdgcn_s_setprio(0);
  RB_STAMP(1);

  const int b_off = (lane >> 5) * XW + (lane & 31) + (PA - a.pad);
  float bcur[4][NB], bnxt[4][NB];
  {
    const float* xt = xs + b_off;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bcur[j][nb] = xt[(2 * j) * XW + nb * 32];
  }

#pragma unroll 1
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const float* xt = xs + (chunk & 3) * BUF + b_off;
    const float* xn = xs + ((chunk + 1) & 3) * BUF + b_off;
    // (the stores and requests past the last chunk run too — clamped addresses, buffers nobody reads: a load or an LDS
    // store behind a branch makes the compiler's s_waitcnt insertion assume either path at every join, i.e. drain)
    const int sbuf = (chunk + 2) & 3;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      // side work of the chunk, between two steps' MFMAs: the tile of chunk + 2 into the ring, the request for
      // chunk + 3, the chunk's barrier
      if (s < NE) {
        if (!(RB_ABL & 1)) lstore1(sbuf, chunk + 2, s);
      } else if (s == NE) {
        if (!(RB_ABL & 1)) gload(chunk + 3);
      } else if (s == NE + 1) {
        if (!(RB_ABL & 4)) __syncthreads();
      }
      if (!(RB_ABL & 2)) ar[(s + RD - 1) % RD] = wq[a_index(chunk, s + RD - 1)];
      if (!(RB_ABL & 8)) {
        const int oi = (s + 1 < S) ? (s + 1) / K : 0;
        const int k = (s + 1 < S) ? (s + 1) - oi * K : 0;
        const float* bp = ((s + 1 < S) ? xt : xn) + (oi * 8) * XW + k * a.dil;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bnxt[j][nb] = bp[(2 * j) * XW + nb * 32];
      }
      // one accumulator (NB = 1): every MFMA depends on the previous one and a filler between two dependent MFMAs costs
      // ~40 cycles (MI355X_MICROARCH.md) — the fillers go first, the MFMAs stay back to back
      if constexpr (NB == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 af = ar[s % RD];
        const float av = (j == 0) ? af.x : (j == 1) ? af.y : (j == 2) ? af.z : af.w;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bcur[j][nb], acc[nb], 0, 0, 0);
      }
      // two accumulators: every filler (the fragment load, the LDS reads of the next step) in the shadow of an MFMA
      if constexpr (NB >= 2) {
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          else if (i < 1 + 2 * NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(RB_ABL & 8)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bcur[j][nb] = bnxt[j][nb];
      }
    }
    if constexpr (S % RD != 0 && !(RB_ABL & 2)) {  // re-base the ring: the next chunk's steps 0 .. RD-2 sit in slots (S + i) % RD
      float4 rr[RD - 1];
#pragma unroll
      for (int i = 0; i < RD - 1; ++i) rr[i] = ar[(S + i) % RD];
#pragma unroll
      for (int i = 0; i < RD - 1; ++i) ar[i] = rr[i];
    }
  }
  RB_STAMP(2);
  if (RB_PRIO) __builtin_amdgcn_s_setprio(RB_PRIO);

  if constexpr (EPI == EPI_UPSAMPLE) {
    // conv_tile's UPSAMPLE epilogue: virtual row co * u + r of GEMM column q -> y[co][q u + r - up_pad]
    const int col = lane & 31, rbase = 4 * (lane >> 5);
    float bb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = a.bias ? a.bias[mt0 * 32 + (r & 3) + 8 * (r >> 2) + rbase] : 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int q = t0 + nb * 32 + col;
      if (q >= n_len) continue;
      if ((RB_ABL & 32) && q != 0) continue;  // (probe builds: the scatter stores ablated)
      if ((a.up & 3) == 0 && (a.up_pad & 3) == 0 && (a.y_ld & 3) == 0) {
        // registers 4g .. 4g+3 of a lane = 4 consecutive phases of ONE output channel = 4 consecutive, 16-byte aligned samples
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int row0 = mt0 * 32 + 8 * g4 + rbase;
          if (row0 >= a.rows) continue;
          const int co = row0 / a.up;
          const int n0 = q * a.up + (row0 - co * a.up) - a.up_pad;
          float* dst = a.y + (long long)b * a.y_bs + (long long)co * a.y_ld + n0;
          if (n0 >= 0 && n0 + 3 < Lout) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[nb][4 * g4] + bb[4 * g4], acc[nb][4 * g4 + 1] + bb[4 * g4 + 1],
                                                          acc[nb][4 * g4 + 2] + bb[4 * g4 + 2], acc[nb][4 * g4 + 3] + bb[4 * g4 + 3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n0 + e >= 0 && n0 + e < Lout) dst[e] = acc[nb][4 * g4 + e] + bb[4 * g4 + e];
          }
        }
        continue;
      }
      if (a.up == 2) {  // registers (r, r + 1), r even = the two phases of one channel: one (4-byte aligned) 8-byte store per pair
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int row = mt0 * 32 + (r & 3) + 8 * (r >> 2) + rbase;  // even
          if (row >= a.rows) continue;
          const int co = row >> 1;
          const int n0 = q * 2 - a.up_pad;
          float* dst = a.y + (long long)b * a.y_bs + (long long)co * a.y_ld + n0;
          const float v0 = acc[nb][r] + bb[r], v1 = acc[nb][r + 1] + bb[r + 1];
          if (n0 >= 0 && n0 + 1 < Lout && row + 1 < a.rows) {
            typedef float up_float2 __attribute__((ext_vector_type(2), aligned(4)));
            *reinterpret_cast<up_float2*>(dst) = up_float2{v0, v1};
          } else {
            if (n0 >= 0 && n0 < Lout) dst[0] = v0;
            if (row + 1 < a.rows && n0 + 1 >= 0 && n0 + 1 < Lout) dst[1] = v1;
          }
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt0 * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (row >= a.rows) continue;
        const int co = row / a.up;
        const int n = q * a.up + (row - co * a.up) - a.up_pad;
        if (n < 0 || n >= Lout) continue;
        a.y[(long long)b * a.y_bs + (long long)co * a.y_ld + n] = acc[nb][r] + bb[r];
      }
    }
  } else {
  // epilogue (conv_tile's LINEAR epilogue without k-groups): y = (acc + bias) [+ res]
  // C/D map of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
  const int row0 = mt0 * 32 + rbase;  // this lane's rows: row0 + (r & 3) + 8 (r >> 2)
  float* yb = a.y + (long long)b * a.y_bs;
  const float* rb = a.res ? a.res + (long long)b * a.y_bs : nullptr;
  const bool interior = tile_y * 128 + 128 <= a.rows && t0 + T_T <= Lout;  // uniform: no masks, no clamps
  float bb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bb[r] = (RB_ABL & 16) ? 0.01f : a.bias[row0 + (r & 3) + 8 * (r >> 2)];  // packed bias: padded to whole m-tiles, never null here
  if (interior && NB % 2 == 0 && RB_WIDE_STORES && (a.y_ld & 3) == 0) {
    // Interior tiles: the accumulator blocks go through LDS once ([32 rows][64 columns] per wave, over the dead ring) so that
    // a lane holds FOUR consecutive columns: 8 dwordx4 residual loads + 8 dwordx4 stores per lane instead of 32 + 32 dword ones
    // (the epilogue of such a tile is store-ISSUE-bound: MI355X_MICROARCH.md).  Same arithmetic: (acc + bias) + residual.
    __syncthreads();  // every wave has issued its last operand read of the ring
    float* tw = xs + wm * (32 * 64);
    const int rbase0 = mt0 * 32;
#pragma unroll
    for (int np = 0; np < NB / 2; ++np) {  // 64 columns at a time (NB = 4: the wave's 8 KB twice; its LDS operations execute in order)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + rbase) * 64 + nb * 32 + col] = acc[2 * np + nb][r] + bb[r];
    __builtin_amdgcn_wave_barrier();  // no instruction: LDS operations of one wave execute in order, its reads below see its writes
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 v[4], rv[4];
      int off[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * (4 * h + i);
        const int row = idx >> 4, c4 = idx & 15;
        off[i] = (rbase0 + row) * a.y_ld + t0 + 64 * np + 4 * c4;
        v[i] = *reinterpret_cast<const float4*>(tw + row * 64 + 4 * c4);
      }
      if (rb && !(RB_ABL & 16)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rv[i] = *reinterpret_cast<const float4*>(rb + off[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i].x += rv[i].x;
          v[i].y += rv[i].y;
          v[i].z += rv[i].z;
          v[i].w += rv[i].w;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(yb + off[i]) = v[i];
    }
    if (np + 1 < NB / 2) __builtin_amdgcn_wave_barrier();
    }
  } else if (interior) {
    const int o0 = row0 * a.y_ld + t0 + col;  // a plane of one batch row is < 2^31 floats (conv_tile indexes it the same way)
    if (rb && !(RB_ABL & 16)) {  // one batch of residual loads per column block (the first goes out together with the bias loads)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = rb[o0 + ((r & 3) + 8 * (r >> 2)) * a.y_ld + nb * 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) yb[o0 + ((r & 3) + 8 * (r >> 2)) * a.y_ld + nb * 32] = (acc[nb][r] + bb[r]) + rv[r];
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) yb[o0 + ((r & 3) + 8 * (r >> 2)) * a.y_ld + nb * 32] = acc[nb][r] + bb[r];
    }
  } else {  // edge tiles (the row's last time tile, rows past the tensor): clamped loads, masked stores, eight registers at a time
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int t = t0 + nb * 32 + col;
      const bool tok = t < Lout;
      const int tc = tok ? t : Lout - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
        int off[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * h + q;
          const int row = row0 + (r & 3) + 8 * (r >> 2);
          off[q] = (row < a.rows ? row : a.rows - 1) * a.y_ld + tc;
          v[q] = acc[nb][r] + bb[r];
        }
        if (rb) {
          float rv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) rv[q] = rb[off[q]];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] += rv[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = 8 * h + q;
          if (tok && row0 + (r & 3) + 8 * (r >> 2) < a.rows) yb[off[q]] = v[q];
        }
      }
    }
  }
  }
  RB_STAMP(3);
}

// one conv per launch on the same tile (the polyphase upsamplers): grid = (time tiles, 128-row tiles, batch rows)
template <int K, int HALO, int EPI, bool MRF, int CI_C = 16>
__global__ __launch_bounds__(256, CI_C == 16 ? 4 : 3) void rb_conv_kernel(const ConvArgs a) {
  __shared__ float xs[rb_lds_floats<HALO, 2, CI_C>()];
  int tile_x, tile_y;
  int gx = gridDim.x;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (conv_mfma.h, row_tiles)
    gx = row_tiles(conv_n_len<K, EPI>(a, blockIdx.z), 64);
    if (lin >= gx * (int)gridDim.y) return;
  }
  xcd_tile_lin(lin, gx, gridDim.y, tile_x, tile_y, a.rows_major);
  rb_tile<K, HALO, EPI, MRF, 2, CI_C>(a, tile_x, tile_y, blockIdx.z, xs);
}

// halo of the staged tile per tap count: (K - 1) d for d <= 5, plus the 4-alignment slack of the tile origin
template <int K> struct RbCfg;
template <> struct RbCfg<11> { static constexpr int HALO = 56; };
template <> struct RbCfg<7> { static constexpr int HALO = 36; };
template <> struct RbCfg<3> { static constexpr int HALO = 16; };

// the three MRF chains' same-geometry convs in one launch, longest first (cf. conv_group_kernel)
template <int K0, int K1, int K2, int NB = 2>
__global__ __launch_bounds__(256, NB > 2 ? 3 : 4) void rb_group_kernel(const ConvGroupArgs g) {
  constexpr int H0 = RbCfg<K0>::HALO, H1 = RbCfg<K1>::HALO, H2 = RbCfg<K2>::HALO;
  constexpr int L0 = rb_lds_floats<H0, NB>(), L1 = rb_lds_floats<H1, NB>(), L2 = rb_lds_floats<H2, NB>();
  __shared__ float xs[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  const bool ragged = gridDim.z > 1;
  int tx, ty;
  CONV_WG_STAMP(lin, 0);
  int m, l;  // member and tile of this workgroup
  if (g.nseg) {  // the host's dispatch order (group_snake_order)
    int sg = 0;
    while (sg + 1 < g.nseg && lin >= g.seg_off[sg + 1]) ++sg;
    m = g.seg_m[sg];
    l = g.seg_first[sg] + (lin - g.seg_off[sg]);
  } else {
    m = lin < g.off[1] ? 0 : lin < g.off[2] ? 1 : 2;
    l = lin - g.off[m];
  }
  if (m == 0) {
    const int gx = ragged ? row_tiles(conv_n_len<K0, EPI_LINEAR>(g.c[0], b), 32 * NB) : g.gx[0];
    if (l >= gx * g.gy[0]) return;
    xcd_tile_lin(l, gx, g.gy[0], tx, ty);
    rb_tile<K0, H0, EPI_LINEAR, false, NB>(g.c[0], tx, ty, b, xs);
  } else if (m == 1) {
    const int gx = ragged ? row_tiles(conv_n_len<K1, EPI_LINEAR>(g.c[1], b), 32 * NB) : g.gx[1];
    if (l >= gx * g.gy[1]) return;
    xcd_tile_lin(l, gx, g.gy[1], tx, ty);
    rb_tile<K1, H1, EPI_LINEAR, false, NB>(g.c[1], tx, ty, b, xs);
  } else {
    const int gx = ragged ? row_tiles(conv_n_len<K2, EPI_LINEAR>(g.c[2], b), 32 * NB) : g.gx[2];
    if (l >= gx * g.gy[2]) return;
    xcd_tile_lin(l, gx, g.gy[2], tx, ty);
    rb_tile<K2, H2, EPI_LINEAR, false, NB>(g.c[2], tx, ty, b, xs);
  }
  CONV_WG_STAMP(lin, 1);
}

}  // namespace mi355tts
