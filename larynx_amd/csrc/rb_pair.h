// One ResBlock1 step of HiFi-GAN — y = x + conv2_{K,1}(lrelu(conv1_{K,d}(lrelu(x)))), hifi_gan/models.py:91-98 — for the
// 64- / 32-channel stages, on FOUR waves without a k-split (round 4).
//
// resblock_pair.h runs the same fusion on 8 waves = 4 time-waves x 2 k-groups: the two k-groups' partial tiles meet in a
// reduce-scatter through LDS after EACH conv (two exchanges, seven barriers per tile), a wave carries 2 m-blocks x 1
// column block at C = 64 (one A-fragment register per MFMA — the fragment stream the round-4 ablations found to be the
// largest single cost of such a tile), and a k = 3 phase is 48-96 MFMAs per wave between two synchronisations.  Here a
// workgroup is 4 waves, each one m-block x TWO column blocks over ALL input channels (half the fragment bytes per MFMA at
// C = 64, twice the MFMAs per wave and phase, no partial tiles at all):
//   C = 64: waves = 2 m-blocks x 2 time groups, 128 conv1 columns per workgroup, 47 KB of LDS, three workgroups per CU;
//   C = 32: waves = 4 time groups, 256 conv1 columns, 40 KB, three workgroups per CU (by registers).
// Per tile: stage lrelu(x) -> barrier -> conv1 -> barrier (x dead) -> park lrelu(conv1 + b1) OVER the x tile -> barrier ->
// conv2 -> + b2 + x -> y: three barriers.  (Requesting conv2's bias and residual before its MFMA phase — RBP_PREFETCH64 / 32 —
// measured equal: profiles/r04_ab14.txt.)
// Same tiles (T2 = T1 - (K - 1) output columns) and arguments as resblock_pair.h; the summation order differs (no
// k-groups), so every schedule of a model takes the same one of the two kernels (option "rb_pair").
#pragma once
#include "conv_mfma.h"
#include "resblock_pair.h"

#ifndef RBP_PREFETCH64  // conv2's bias + residual requested BEFORE its MFMA phase (48 registers through the phase), C = 64 / 32
#define RBP_PREFETCH64 0
#endif
#ifndef RBP_PREFETCH32
#define RBP_PREFETCH32 0
#endif
#ifndef RBP_WIDE_STORES  // interior tiles: accumulators transposed through LDS, 16-byte residual loads and stores (rb_conv.h)
#define RBP_WIDE_STORES 1
#endif
#ifndef RBP_LB32  // waves per SIMD the C = 32 kernel is compiled for: 3 (132 VGPRs, no spills; measured 129.8 us per grouped
#define RBP_LB32 3  // launch) or 4 (four 40 KB workgroups per CU, 3 spilled registers: 132.1 us)
#endif

namespace mi355tts {

template <int K, int CB>
struct RbPairGeom {
  static constexpr int C = CB * 32;
  static constexpr int TG = 4 / CB;                                  // time groups of waves
  static constexpr int T1 = TG * 64;                                 // conv1 columns per workgroup (2 blocks of 32 per wave)
  static constexpr int XW = (T1 + (K - 1) * PAIR_DMAX + 4 + 3) & ~3;  // staged x row: tile + conv1 halo + alignment slack
  static constexpr int TW = (T1 + K + 3) & ~3;                       // parked conv1 row (+ slack for the masked tail reads)
  static constexpr int LDS = C * XW;                                  // floats; the parked tile (C x TW) aliases the x tile
  static_assert(TW <= XW, "the parked tile must fit over the x tile");
};

// one MFMA phase of a wave: acc[nb] += sum over (octet, tap) of A-fragment x B-column; bt = this lane's first B element
template <int K, int NOCT>
__device__ __forceinline__ void rbp_phase(floatx16 (&acc)[2], const float4* wq, const float* bt, int RS, int dil) {
  constexpr int S = NOCT * K;
  auto a_index = [&](int q) -> long long { return (long long)(q < S ? q : S - 1) * 64; };  // (octet, tap) pairs are consecutive
  float4 ar[3];
  ar[0] = wq[a_index(0)];
  ar[1] = wq[a_index(1)];
  float bcur[4][2], bnxt[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bcur[j][nb] = bt[(2 * j) * RS + nb * 32];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    ar[(s + 2) % 3] = wq[a_index(s + 2)];
    if (s + 1 < S) {
      const int oi = (s + 1) / K;
      const int k = (s + 1) - oi * K;
      const float* bp = bt + (oi * 8) * RS + k * dil;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bnxt[j][nb] = bp[(2 * j) * RS + nb * 32];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 af = ar[s % 3];
      const float av = (j == 0) ? af.x : (j == 1) ? af.y : (j == 2) ? af.z : af.w;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bcur[j][nb], acc[nb], 0, 0, 0);
    }
    // every filler (the fragment load, the LDS reads of the next step) in the shadow of an MFMA
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (i < 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      else if (i < 5) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < S) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) bcur[j][nb] = bnxt[j][nb];
    }
  }
}

// One workgroup's tile: output columns [tile_x * T2, +T2) of batch row b.  xs = RbPairGeom::LDS floats of LDS.
template <int K, int CB>
__device__ __forceinline__ void rbp_tile(const PairArgs& a, const int tile_x, const int b, float* __restrict__ xs) {
  using G = RbPairGeom<K, CB>;
  constexpr int C = G::C, T1 = G::T1, XW = G::XW, TW = G::TW;
  constexpr int P2 = (K - 1) / 2;  // conv2 "same" padding
  constexpr int T2 = T1 - 2 * P2;  // output columns per workgroup
  constexpr int NOCT = C / 8;
  constexpr int NF4 = C * (XW / 4), NE = (NF4 + 255) / 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mb = wave % CB, tg = wave / CB;
  const int col = lane & 31, half = lane >> 5, rbase = 4 * half;
  const int wcol0 = tg * 64;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int j0 = tile_x * T2;  // first output column of this workgroup
  if (j0 >= L) return;
  const int gt0 = j0 - P2;             // global column of parked-tile column 0
  const int p1 = (K - 1) * a.dil / 2;  // conv1 "same" padding
  const int gx0 = (gt0 - p1) & ~3;     // 4-aligned global column of staged-tile column 0
  const int shift = (gt0 - p1) - gx0;
  const float* xb = a.x + (long long)b * a.bs;
  const float slope = a.slope;

  // first fragments of conv1 and conv1's bias: requested with the activation tile
  const float4* wq1 = reinterpret_cast<const float4*>(a.w1) + (long long)mb * a.noct * K * 64 + lane;
  const float4* wq2 = reinterpret_cast<const float4*>(a.w2) + (long long)mb * a.noct * K * 64 + lane;
  // ---- phase 0: stage lrelu(x) for all C channels, 16-byte loads, branch-free, in batches of NEB loads per thread
  {
    const int ld_last4 = a.ld - 4;
    constexpr int NEB = NE > 6 ? (NE + 1) / 2 : NE;
#pragma unroll
    for (int i0 = 0; i0 < NE; i0 += NEB) {
      float4 pre[NEB];
#pragma unroll
      for (int i = 0; i < NEB; ++i) {
        const int e = tid + 256 * (i0 + i);
        const int row = e / (XW / 4), f = e - row * (XW / 4);
        const int c0 = gx0 + 4 * f;
        pre[i] = *reinterpret_cast<const float4*>(xb + (row < C ? row : C - 1) * a.ld + (c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0)));
      }
#pragma unroll
      for (int i = 0; i < NEB; ++i) {
        const int e = tid + 256 * (i0 + i);
        const int row = e / (XW / 4), f = e - row * (XW / 4);
        const int c0 = gx0 + 4 * f;
        float4 v = pre[i];
        v.x = (c0 >= 0 && c0 < L) ? v.x : 0.f;
        v.y = (c0 + 1 >= 0 && c0 + 1 < L) ? v.y : 0.f;
        v.z = (c0 + 2 >= 0 && c0 + 2 < L) ? v.z : 0.f;
        v.w = (c0 + 3 >= 0 && c0 + 3 < L) ? v.w : 0.f;
        v.x = v.x > 0.f ? v.x : v.x * slope;
        v.y = v.y > 0.f ? v.y : v.y * slope;
        v.z = v.z > 0.f ? v.z : v.z * slope;
        v.w = v.w > 0.f ? v.w : v.w * slope;
        if (e < NF4) reinterpret_cast<float4*>(xs)[e] = v;
        (void)row;
      }
    }
  }
  __syncthreads();

  floatx16 acc[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  };

  // ---- phase 1: conv1 (dilation d) on T1 columns
  zero_acc();
  rbp_phase<K, NOCT>(acc, wq1, xs + half * XW + shift + wcol0 + col, XW, a.dil);
  {
    float bb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = a.b1[mb * 32 + (r & 3) + 8 * (r >> 2) + rbase];
    __syncthreads();  // every wave has read its last x column: the parked tile goes over the x tile
    // park lrelu(conv1 + bias); columns outside the sequence are conv2's ZERO padding
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int jj = wcol0 + nb * 32 + col;
      const int g = gt0 + jj;
      const bool inside = g >= 0 && g < L;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mb * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        float v = acc[nb][r] + bb[r];
        v = v > 0.f ? v : v * slope;
        xs[row * TW + jj] = inside ? v : 0.f;
      }
    }
  }
  __syncthreads();

  // ---- phase 2: conv2 (dilation 1) on the parked tile
  float bb2[16], rv[2][16];
  int gcs[2];
  bool toks[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int jj = wcol0 + nb * 32 + col;
    const int g = j0 + jj;
    toks[nb] = jj < T2 && g < L;
    gcs[nb] = g < L ? g : L - 1;
  }
  auto epi_loads = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) bb2[r] = a.b2[mb * 32 + (r & 3) + 8 * (r >> 2) + rbase];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) rv[nb][r] = xb[(mb * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld + gcs[nb]];
  };
  constexpr bool PREFETCH = CB == 2 ? (RBP_PREFETCH64 != 0) : (RBP_PREFETCH32 != 0);
  const bool interior = RBP_WIDE_STORES && !PREFETCH && j0 + T1 <= L;  // uniform: every column of the tile is inside the sequence
  if (PREFETCH) epi_loads();
  zero_acc();
  rbp_phase<K, NOCT>(acc, wq2, xs + half * TW + wcol0 + col, TW, 1);
  if (interior) {
    // The accumulator blocks go through LDS once ([32 rows][64 columns] per wave, over the dead parked tile) so that a lane
    // holds FOUR consecutive columns: 8 16-byte residual loads + 8 16-byte stores per lane instead of 32 + 32 dword ones.  A
    // tile starts at column tile_x * T2, so these are dword-aligned 16-byte accesses (global memory takes them).  Same
    // arithmetic, same order: ((acc + bias) + x) * alpha [+ y].
    typedef float rbp_f4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
    for (int r = 0; r < 16; ++r) bb2[r] = a.b2[mb * 32 + (r & 3) + 8 * (r >> 2) + rbase];
    __syncthreads();  // every wave has read its last parked column
    float* tw = xs + wave * (32 * 64);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + rbase) * 64 + nb * 32 + col] = acc[nb][r] + bb2[r];
    __builtin_amdgcn_wave_barrier();  // no instruction: a wave's LDS operations execute in order
    float* yb = a.y + (long long)b * a.bs;
    const float alpha = a.alpha;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 v[4];
      rbp_f4 rx[4];
      int off[4], jjs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * (4 * h + i);
        const int row = idx >> 4, c4 = idx & 15;
        jjs[i] = wcol0 + 4 * c4;
        off[i] = (mb * 32 + row) * a.ld + j0 + jjs[i];
        v[i] = *reinterpret_cast<const float4*>(tw + row * 64 + 4 * c4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) rx[i] = *reinterpret_cast<const rbp_f4*>(xb + off[i]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i].x = (v[i].x + rx[i].x) * alpha;
        v[i].y = (v[i].y + rx[i].y) * alpha;
        v[i].z = (v[i].z + rx[i].z) * alpha;
        v[i].w = (v[i].w + rx[i].w) * alpha;
      }
      if (a.accum) {
        rbp_f4 ov[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = *reinterpret_cast<const rbp_f4*>(yb + off[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i].x += ov[i].x;
          v[i].y += ov[i].y;
          v[i].z += ov[i].z;
          v[i].w += ov[i].w;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (jjs[i] + 3 < T2) {
          *reinterpret_cast<rbp_f4*>(yb + off[i]) = rbp_f4{v[i].x, v[i].y, v[i].z, v[i].w};
        } else {  // the tile's last columns (T2 is not a multiple of four)
          if (jjs[i] < T2) yb[off[i]] = v[i].x;
          if (jjs[i] + 1 < T2) yb[off[i] + 1] = v[i].y;
          if (jjs[i] + 2 < T2) yb[off[i] + 2] = v[i].z;
        }
      }
    }
    return;
  }
  if (!PREFETCH) epi_loads();
  // ---- epilogue: + bias + residual
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    float* yb = a.y + (long long)b * a.bs + gcs[nb];
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (acc[nb][r] + bb2[r] + rv[nb][r]) * a.alpha;
    if (a.accum) {
      float ov[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ov[r] = yb[(mb * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += ov[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (toks[nb]) yb[(mb * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld] = v[r];
  }
}

template <int K, int CB>
__global__ __launch_bounds__(256, CB == 2 ? 3 : RBP_LB32) void rb_pair_kernel(const PairArgs a) {
  __shared__ float xs[RbPairGeom<K, CB>::LDS];
  int tile_x, tile_y;
  int gx = gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (conv_mfma.h, row_tiles)
    gx = row_tiles(a.len ? a.len[blockIdx.z] * a.len_mul : a.len_const, RbPairGeom<K, CB>::T1 - (K - 1));
    if ((int)blockIdx.x >= gx) return;
  }
  xcd_tile_lin(blockIdx.x, gx, 1, tile_x, tile_y);  // neighbouring tiles share their halo through one XCD's L2
  rbp_tile<K, CB>(a, tile_x, blockIdx.z, xs);
}

// the fused steps of the three MRF chains of a stage in ONE launch, longest member first (cf. pair_group_kernel)
template <int K0, int K1, int K2, int CB>
__global__ __launch_bounds__(256, CB == 2 ? 3 : RBP_LB32) void rb_pair_group_kernel(const PairGroupArgs g) {
  constexpr int L0 = RbPairGeom<K0, CB>::LDS, L1 = RbPairGeom<K1, CB>::LDS, L2 = RbPairGeom<K2, CB>::LDS;
  __shared__ float xs[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  auto tiles = [&](const PairArgs& p, int gx_grid, int t2) { return gridDim.z > 1 ? row_tiles(p.len ? p.len[b] * p.len_mul : p.len_const, t2) : gx_grid; };
  constexpr int T1 = RbPairGeom<K0, CB>::T1;
  int tx, ty;
  CONV_WG_STAMP(lin, 0);
  if (lin < g.off[1]) {
    const int gx = tiles(g.p[0], g.gx[0], T1 - (K0 - 1));
    if (lin >= gx) return;
    xcd_tile_lin(lin, gx, 1, tx, ty);
    rbp_tile<K0, CB>(g.p[0], tx, b, xs);
  } else if (lin < g.off[2]) {
    const int l = lin - g.off[1];
    const int gx = tiles(g.p[1], g.gx[1], T1 - (K1 - 1));
    if (l >= gx) return;
    xcd_tile_lin(l, gx, 1, tx, ty);
    rbp_tile<K1, CB>(g.p[1], tx, b, xs);
  } else {
    const int l = lin - g.off[2];
    const int gx = tiles(g.p[2], g.gx[2], T1 - (K2 - 1));
    if (l >= gx) return;
    xcd_tile_lin(l, gx, 1, tx, ty);
    rbp_tile<K2, CB>(g.p[2], tx, b, xs);
  }
  CONV_WG_STAMP(lin, 1);
}

}  // namespace mi355tts
