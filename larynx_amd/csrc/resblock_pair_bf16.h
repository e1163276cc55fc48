// One ResBlock1 step of HiFi-GAN fused into a single kernel on the bf16 matrix cores with split
// operands — the `half`-mode counterpart of resblock_pair.h for the narrow stages (C = 32 / 64):
//
//   y = x + conv2_{K,d=1}( lrelu( conv1_{K,d}( lrelu(x) ) ) )      hifi_gan/models.py:91-98
//
// Un-fused, the two convs of a step at C = 32 / 64 move 100 MB per 2-7 GFLOP (x in, t out; t in,
// x in again as the residual, y out) and the split-bf16 kernel runs them at 1.2-1.4x their HBM
// floor.  Fused, the intermediate never leaves the CU: lrelu(x) is staged ONCE for all C channels
// as bf16 hi/lo planes ([octet][column] groups of 8 channels, conv_bf16.h's layout), conv1 is
// evaluated on T1 = T2 + (K-1) columns (the halo conv2 needs is recomputed), its activated output
// is split again and parked in LDS over the dead input tile, conv2 consumes it from there; bias +
// f32 residual in the epilogue.  40 MB per pair instead of 100.
//
// One workgroup = WM x WN waves (WM = C / 32 row groups x WN time-waves); a wave owns one 32-row
// m-block x NB column blocks over ALL input channels (no k-split).  MFMA / operand pipeline as in
// conv_bf16.h: weight fragments two steps ahead, LDS fragments one step ahead, pinned behind the
// MFMAs with sched_group_barrier.  Measured on 'high' at batch 1 (grouped launches, same box): C = 64 87 µs fused vs
// 2 x 53 un-fused, C = 32 53 vs 2 x 44; 4 time-waves x 2 column blocks beats 2 x 4 (107 / 77 µs) and 8 x 1 (98 / 59).
#pragma once
#include <hip/hip_runtime.h>

#include "conv_bf16.h"
#include "resblock_pair.h"

namespace mi355tts {

template <int K, int WM, int WN, int NB>
struct PairBf16Geom {
  static constexpr int C = WM * 32;
  static constexpr int T1 = WN * NB * 32;                              // conv1 columns per workgroup
  static constexpr int HALO = ((K - 1) * PAIR_DMAX + 3 + 3) & ~3;      // conv1 halo + alignment slack
  static constexpr int XW = T1 + HALO;                                 // staged columns per octet row
  static constexpr int TW = (T1 + K - 1 + 3) & ~3;                     // parked conv1 columns per octet row
  static constexpr int UNITS = 2 * (C / 8) * XW;                       // LDS, in uint4 units (two planes)
  static_assert(TW <= XW, "the parked tile aliases the staged input tile");
};

// One MFMA phase: acc[nb] += sum over (16-channel slab, tap) of A-fragment x B-fragment.
// wq = this wave's packed bf16 weights (+ lane); bt = this lane's first B unit (plane hi, octet `ohalf`, its first
// column); RS = units per octet row, PL = units per plane; tap k is k*dil columns to the right.
template <int K, int NB, int NSLAB, int TERMS>
__device__ __forceinline__ void pair_bf16_phase(floatx16 (&acc)[NB], const uint4* __restrict__ wq, const uint4* __restrict__ bt, const int RS,
                                                const int PL, const int dil) {
  constexpr int S = NSLAB * K;
  constexpr int AD = 2;
  auto a_off = [&](int q) -> int { return (q < S ? q : S - 1) * 128; };  // steps are (slab, tap) in packed order
  uint4 Ah[AD + 1], Al[AD + 1];
  uint4 B0h[NB], B0l[NB], B1h[NB], B1l[NB];
#pragma unroll
  for (int d = 0; d < AD; ++d) {
    Ah[d] = wq[a_off(d)];
    if (TERMS == 3) Al[d] = wq[a_off(d) + 64];
  }
  auto bread = [&](int q, uint4* bh, uint4* bl) {
    const int s = q / K, k = q - s * K;
    const uint4* bp = bt + (2 * s) * RS + k * dil;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      bh[nb] = bp[nb * 32];
      if (TERMS == 3) bl[nb] = bp[PL + nb * 32];
    }
  };
  bread(0, B0h, B0l);
  __builtin_amdgcn_sched_barrier(0);
  constexpr int NMF = TERMS * NB;
#pragma unroll
  for (int q = 0; q < S; ++q) {
    Ah[AD] = wq[a_off(q + AD)];
    if (TERMS == 3) Al[AD] = wq[a_off(q + AD) + 64];
    if (q + 1 < S) bread(q + 1, B1h, B1l);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf16(Ah[0], B0h[nb], acc[nb]);
    if (TERMS == 3) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf16(Ah[0], B0l[nb], acc[nb]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf16(Al[0], B0h[nb], acc[nb]);
    }
    {
      constexpr int NV = TERMS == 3 ? 2 : 1, ND = (TERMS == 3 ? 2 : 1) * NB;
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        else if (q + 1 < S) {
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
    for (int d = 0; d < AD; ++d) {
      Ah[d] = Ah[d + 1];
      if (TERMS == 3) Al[d] = Al[d + 1];
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      B0h[nb] = B1h[nb];
      if (TERMS == 3) B0l[nb] = B1l[nb];
    }
  }
}

// One workgroup's tile: output columns [tile_x * T2, +T2) of batch row b.  xs = PairBf16Geom::UNITS uint4 of LDS.
template <int K, int WM, int WN, int NB, int TERMS>
__device__ __forceinline__ void pair_bf16_tile(const PairArgs& a, const int tile_x, const int b, uint4* __restrict__ xs) {
  using G = PairBf16Geom<K, WM, WN, NB>;
  constexpr int C = G::C, T1 = G::T1, XW = G::XW, TW = G::TW;
  constexpr int P2 = (K - 1) / 2;   // conv2 "same" padding
  constexpr int T2 = T1 - 2 * P2;   // output columns per workgroup
  constexpr int NT = 64 * WM * WN;
  constexpr int OCT = C / 8, XQ = XW / 4;
  constexpr int XPL = OCT * XW, TPL = OCT * TW;  // units per plane of the staged / parked tile
  constexpr int NUNITS = OCT * XQ;
  constexpr int NU = (NUNITS + NT - 1) / NT;
  constexpr int NSLAB = C / 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;
  const int wm = wave / WN;
  const int col = lane & 31, half = lane >> 5;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int j0 = tile_x * T2;  // first output column of this workgroup
  if (j0 >= L) return;
  const int gt0 = j0 - P2;                  // global column of parked-tile column 0
  const int p1 = (K - 1) * a.dil / 2;       // conv1 "same" padding
  const int gx0 = (gt0 - p1) & ~3;          // 4-aligned global column of staged-tile column 0
  const int shift = (gt0 - p1) - gx0;
  const float* xb = a.x + (long long)b * a.bs;
  const float slope = a.slope;

  // ---- phase 0: stage lrelu(x), all C channels, as bf16 hi/lo planes; unit = 8 channels x 4 columns
  {
    const int ld_last4 = a.ld - 4;
    float4 pre[NU][8];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = tid + NT * i;
      const int o = u < NUNITS ? u / XQ : 0, q = u < NUNITS ? u - (u / XQ) * XQ : 0;
      const int c0 = gx0 + 4 * q;
      const int cc = c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) pre[i][j] = *reinterpret_cast<const float4*>(xb + (long long)(8 * o + j) * a.ld + cc);
    }
    const bool inside = gx0 >= 0 && gx0 + XW <= L && slope >= 0.f && slope <= 1.f;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = tid + NT * i;
      if (u >= NUNITS) continue;
      const int o = u / XQ, q = u - o * XQ;
      const int c0 = gx0 + 4 * q;
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
          const float4 p = pre[i][j];
          v[j][0] = (c0 >= 0 && c0 < L) ? p.x : 0.f;
          v[j][1] = (c0 + 1 >= 0 && c0 + 1 < L) ? p.y : 0.f;
          v[j][2] = (c0 + 2 >= 0 && c0 + 2 < L) ? p.z : 0.f;
          v[j][3] = (c0 + 3 >= 0 && c0 + 3 < L) ? p.w : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] > 0.f ? v[j][e] : v[j][e] * slope;
        }
      }
      uint4* dst = xs + o * XW + 4 * q;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint4 hi, lo;
        split_bf16(v[0][e], v[1][e], hi.x, lo.x);
        split_bf16(v[2][e], v[3][e], hi.y, lo.y);
        split_bf16(v[4][e], v[5][e], hi.z, lo.z);
        split_bf16(v[6][e], v[7][e], hi.w, lo.w);
        dst[e] = hi;
        dst[XPL + e] = lo;
      }
    }
  }
  __syncthreads();

  const uint4* wq1 = reinterpret_cast<const uint4*>(a.w1h) + (long long)wm * a.nslab * K * 128 + lane;
  const uint4* wq2 = reinterpret_cast<const uint4*>(a.w2h) + (long long)wm * a.nslab * K * 128 + lane;
  floatx16 acc[NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  };
  const int rbase = 4 * half;

  // ---- phase 1: conv1 (dilation d) on T1 columns
  zero_acc();
  pair_bf16_phase<K, NB, NSLAB, TERMS>(acc, wq1, xs + half * XW + shift + wn * (NB * 32) + col, XW, XPL, a.dil);
  __syncthreads();  // every wave is done with the staged input: the parked tile may overwrite it
  {
    // park lrelu(conv1 + bias) as bf16 hi/lo; columns outside the sequence are conv2's ZERO padding.  A lane holds
    // channels 8*o + 4*half + (0..3) of its column for the four octets o of its m-block: one 8-byte store per plane.
    float bb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = a.b1[wm * 32 + (r & 3) + 8 * (r >> 2) + rbase];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int jj = (wn * NB + nb) * 32 + col;
      const int g = gt0 + jj;
      const bool in_seq = g >= 0 && g < L;
#pragma unroll
      for (int og = 0; og < 4; ++og) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = acc[nb][4 * og + e] + bb[4 * og + e];
          t = t > 0.f ? t : t * slope;
          v[e] = in_seq ? t : 0.f;
        }
        uint2 hi, lo;
        split_bf16(v[0], v[1], hi.x, lo.x);
        split_bf16(v[2], v[3], hi.y, lo.y);
        uint2* dst = reinterpret_cast<uint2*>(xs + (wm * 4 + og) * TW + jj) + half;
        *dst = hi;
        *(dst + 2 * TPL) = lo;
      }
    }
  }
  __syncthreads();

  // ---- phase 2: conv2 (dilation 1) on the parked tile
  zero_acc();
  pair_bf16_phase<K, NB, NSLAB, TERMS>(acc, wq2, xs + half * TW + wn * (NB * 32) + col, TW, TPL, 1);

  // ---- epilogue: + bias + f32 residual, batched loads from clamped addresses
  float bb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bb[r] = a.b2[wm * 32 + (r & 3) + 8 * (r >> 2) + rbase];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int jj = (wn * NB + nb) * 32 + col;
    const int g = j0 + jj;
    const bool tok = jj < T2 && g < L;
    const int gc = g < L ? g : L - 1;
    const float* rb = xb + gc;
    float* yb = a.y + (long long)b * a.bs + gc;
    float rv[16], v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = rb[(wm * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (acc[nb][r] + bb[r] + rv[r]) * a.alpha;
    if (a.accum) {
      float ov[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ov[r] = yb[(wm * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += ov[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (tok) yb[(wm * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld] = v[r];
  }
}

// C = 64 (512 threads, 80 KB of LDS): capped at 128 VGPRs so that two workgroups share a CU
template <int K, int WM, int WN, int NB, int TERMS>
__global__ __launch_bounds__(64 * WM * WN, (WM == 2 && WN == 4) ? 4 : 1) void pair_bf16_kernel(const PairArgs a) {
  __shared__ uint4 xs[PairBf16Geom<K, WM, WN, NB>::UNITS];
  int tile_x, tile_y;
  int gx = gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (conv_mfma.h, row_tiles)
    gx = row_tiles(a.len ? a.len[blockIdx.z] * a.len_mul : a.len_const, PairBf16Geom<K, WM, WN, NB>::T1 - (K - 1));
    if ((int)blockIdx.x >= gx) return;
  }
  xcd_tile_lin(blockIdx.x, gx, 1, tile_x, tile_y);  // neighbouring tiles share their halo through one XCD's L2
  pair_bf16_tile<K, WM, WN, NB, TERMS>(a, tile_x, blockIdx.z, xs);
}

// The fused steps of the three MRF chains of a stage in ONE launch (see pair_group_kernel).
template <int K0, int K1, int K2, int WM, int WN, int NB, int TERMS>
__global__ __launch_bounds__(64 * WM * WN, (WM == 2 && WN == 4) ? 4 : 1) void pair_bf16_group_kernel(const PairGroupArgs g) {
  constexpr int U0 = PairBf16Geom<K0, WM, WN, NB>::UNITS, U1 = PairBf16Geom<K1, WM, WN, NB>::UNITS, U2 = PairBf16Geom<K2, WM, WN, NB>::UNITS;
  __shared__ uint4 xs[U0 > U1 ? (U0 > U2 ? U0 : U2) : (U1 > U2 ? U1 : U2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  // ragged batch: a row deals only its own tiles (conv_mfma.h, row_tiles)
  auto tiles = [&](const PairArgs& p, int gx_grid, int t2) { return gridDim.z > 1 ? row_tiles(p.len ? p.len[b] * p.len_mul : p.len_const, t2) : gx_grid; };
  constexpr int T1 = PairBf16Geom<K0, WM, WN, NB>::T1;
  int tx, ty;
  if (lin < g.off[1]) {
    const int gx = tiles(g.p[0], g.gx[0], T1 - (K0 - 1));
    if (lin >= gx) return;
    xcd_tile_lin(lin, gx, 1, tx, ty);
    pair_bf16_tile<K0, WM, WN, NB, TERMS>(g.p[0], tx, b, xs);
  } else if (lin < g.off[2]) {
    const int l = lin - g.off[1];
    const int gx = tiles(g.p[1], g.gx[1], T1 - (K1 - 1));
    if (l >= gx) return;
    xcd_tile_lin(l, gx, 1, tx, ty);
    pair_bf16_tile<K1, WM, WN, NB, TERMS>(g.p[1], tx, b, xs);
  } else {
    const int l = lin - g.off[2];
    const int gx = tiles(g.p[2], g.gx[2], T1 - (K2 - 1));
    if (l >= gx) return;
    xcd_tile_lin(l, gx, 1, tx, ty);
    pair_bf16_tile<K2, WM, WN, NB, TERMS>(g.p[2], tx, b, xs);
  }
}

}  // namespace mi355tts
