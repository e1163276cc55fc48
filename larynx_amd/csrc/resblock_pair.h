// One ResBlock1 step of HiFi-GAN fused into a single kernel, for the narrow stages
// (C = 32 / 64) where both convs of the pair have ALL their channels inside one
// workgroup:
//
//   y = x + conv2_{K,d=1}( lrelu( conv1_{K,d}( lrelu(x) ) ) )      hifi_gan/models.py:91-98
//
// The intermediate never leaves the CU: conv1 is evaluated on T1 = T2 + (K-1)
// columns (the halo conv2 needs is recomputed, (K-1)/T1 extra work), its
// activated output is parked in LDS, conv2 consumes it from there.  Compared with
// two conv_mfma launches this removes one launch, one prologue/epilogue, and the
// write + re-read of a whole [C][L] plane — at C = 32/64 the un-fused pair moves
// 100 MB per 2-7 GFLOP and is bound by cache bandwidth, not by the matrix cores.
//
// Same building blocks as conv_mfma.h: v_mfma_f32_32x32x2_f32 (exact f32), packed
// A-fragment stream from L2 through a 3-deep register ring, activations staged with
// 16-byte loads and the leaky-ReLU applied once on the way into LDS, 4 time-waves x
// 2 k-groups per workgroup, k-groups summed through LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_mfma.h"

namespace mi355tts {

struct PairArgs {
  const float* x;  // input = residual, [B][C][ld]
  float* y;        // output, same geometry (must not alias x)
  long long bs;
  int ld;
  const int* len;  // valid length of row b = len ? len[b] * len_mul : len_const
  int len_mul;
  int len_const;
  const float* w1;  // packed like conv_mfma's weights: [m-tile][octet][tap][64][4]
  const float* b1;
  const float* w2;
  const float* b2;
  int noct;  // packed octets per m-tile (padded)
  int C;
  int dil;  // dilation of conv1 (conv2 has dilation 1)
  float slope;
  float alpha;  // y = [y +] alpha * (...): the serial MRF schedule accumulates here
  int accum;
  // split-bf16 form of the same step (resblock_pair_bf16.h): weights packed by pack_conv_bf16, slabs per m-tile
  const void* w1h = nullptr;
  const void* w2h = nullptr;
  int nslab = 0;
};

constexpr int PAIR_DMAX = 5;  // largest conv1 dilation the staged halo covers

// One MFMA phase: acc[mb][nb] += sum over this k-group's (octet, tap) steps of
// A-fragment x B-column.  `bt` points at this lane's first B element (row `half`, its
// first column); rows are RS floats apart, tap k is k*dil columns to the right.
template <int K, int CB, int NB, int NO>
__device__ __forceinline__ void pair_mfma_phase(floatx16 (&acc)[CB][NB], const float4* const (&wq)[CB], int kg, const float* bt,
                                                int RS, int dil) {
  constexpr int S = NO * K;
  auto a_index = [&](int q) -> long long {
    if (q >= S) q = S - 1;
    const int oi = q / K;
    const int k = q - oi * K;
    return (long long)(((kg + oi * 2) * K + k)) * 64;
  };
  float4 ar[3][CB];
#pragma unroll
  for (int mb = 0; mb < CB; ++mb) {
    ar[0][mb] = wq[mb][a_index(0)];
    ar[1][mb] = wq[mb][a_index(1)];
  }
  float bcur[4][NB], bnxt[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bcur[j][nb] = bt[(kg * 8 + 2 * j) * RS + nb * 32];
#pragma unroll
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int mb = 0; mb < CB; ++mb) ar[(s + 2) % 3][mb] = wq[mb][a_index(s + 2)];
    if (s + 1 < S) {
      const int oi = (s + 1) / K;
      const int k = (s + 1) - oi * K;
      const float* bp = bt + ((kg + oi * 2) * 8) * RS + k * dil;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bnxt[j][nb] = bp[(2 * j) * RS + nb * 32];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int mb = 0; mb < CB; ++mb) {
        const float4 af = ar[s % 3][mb];
        const float av = (j == 0) ? af.x : (j == 1) ? af.y : (j == 2) ? af.z : af.w;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bcur[j][nb], acc[mb][nb], 0, 0, 0);
      }
    }
    if (s + 1 < S) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bcur[j][nb] = bnxt[j][nb];
    }
  }
}

template <int K, int CB, int NB>
struct PairGeom {
  static constexpr int C = CB * 32;
  static constexpr int T1 = 4 * NB * 32;                                 // conv1 columns per workgroup
  static constexpr int XW = (T1 + (K - 1) * PAIR_DMAX + 4 + 3) & ~3;     // staged x row: tile + conv1 halo + alignment slack
  static constexpr int TW = (T1 + K + 3) & ~3;                           // parked conv1 row (+ slack for the masked tail reads)
  static constexpr int XS = C * XW, TS = C * TW;                         // LDS floats
  // The parked conv1 tile lives in the SAME LDS as the x tile, behind the k-group reduction scratch: by the time it is
  // written (after conv1's reduction) nothing reads x from LDS any more.  C = 64: 83 KB -> 52 KB, two workgroups per CU.
#ifdef MI355TTS_PAIR_SPLIT_LDS  // A/B builds: the round-2 layout (x tile and parked tile side by side)
  static constexpr int RED = XS;
#else
  static constexpr int RED = 2 * 4 * 16 * 64;  // the reduce-scatter's exchange scratch: 2 k-groups x 4 time-waves x one block
#endif
  static constexpr int LDS = XS > RED + TS ? XS : RED + TS;
};

// One workgroup's tile: output columns [tile_x * T2, +T2) of batch row b.  xs / ts = PairGeom::XS / TS floats of LDS.
template <int K, int CB, int NB>
__device__ __forceinline__ void pair_tile(const PairArgs& a, const int tile_x, const int b, float* __restrict__ xs, float* __restrict__ ts) {
  constexpr int C = CB * 32;
  constexpr int WN = 4;
  constexpr int T1 = WN * NB * 32;     // conv1 columns per workgroup
  constexpr int P2 = (K - 1) / 2;      // conv2 "same" padding
  constexpr int T2 = T1 - 2 * P2;      // output columns per workgroup
  constexpr int XW = PairGeom<K, CB, NB>::XW;
  constexpr int TW = PairGeom<K, CB, NB>::TW;
  constexpr int NO = (C / 8) / 2;      // octets per k-group
  constexpr int NF4 = C * (XW / 4);
  constexpr int NE = (NF4 + 511) / 512;
  constexpr int RED = PairGeom<K, CB, NB>::RED;
  static_assert(C * XW >= 2 * 4 * 16 * 64, "the exchange scratch must fit in the x tile");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave & 3;
  const int kg = wave >> 2;
  const int col = lane & 31, half = lane >> 5;
  const int rbase = 4 * half;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int j0 = tile_x * T2;  // first output column of this workgroup
  if (j0 >= L) return;
  const int gt0 = j0 - P2;                  // global column of parked-tile column 0
  const int p1 = (K - 1) * a.dil / 2;       // conv1 "same" padding
  const int gx0 = (gt0 - p1) & ~3;          // 4-aligned global column of staged-tile column 0
  const int shift = (gt0 - p1) - gx0;
  const float* xb = a.x + (long long)b * a.bs;
  const float slope = a.slope;

  // ---- phase 0: stage lrelu(x) for all C channels, 16-byte loads, branch-free
  {
    const int ld_last4 = a.ld - 4;
    float4 pre[NE];
    int off[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 512 * i;
      const int row = e / (XW / 4), f = e - row * (XW / 4);
      const int c0 = gx0 + 4 * f;
      off[i] = (row < C ? row : C - 1) * a.ld + (c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0));
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) pre[i] = *reinterpret_cast<const float4*>(xb + off[i]);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 512 * i;
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
  __syncthreads();

  const float4* wq1[CB];
  const float4* wq2[CB];
#pragma unroll
  for (int mb = 0; mb < CB; ++mb) {
    wq1[mb] = reinterpret_cast<const float4*>(a.w1) + (long long)mb * a.noct * K * 64 + lane;
    wq2[mb] = reinterpret_cast<const float4*>(a.w2) + (long long)mb * a.noct * K * 64 + lane;
  }
  floatx16 acc[CB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mb = 0; mb < CB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
  };
  // Reduce-scatter of the two k-groups' partial tiles through LDS (scratch aliases the x tile, dead by then): a wave holds
  // CB x NB = 2 accumulator blocks ("units": the two m-blocks at C = 64, the two column blocks at C = 32); k-group g ends up
  // with the finished unit g — ONE exchange round instead of one reduction round per block, and BOTH k-groups run their
  // half of the epilogue (with a reduce-to-group-0 half the waves sat out the longest memory round trips of the kernel).
  static_assert(CB * NB == 2, "two accumulator blocks per wave: one per k-group");
  // (the unit index is a compile-time constant everywhere: a run-time index into the accumulator array would put it in scratch)
  floatx16& u0 = CB == 2 ? acc[0][0] : acc[0][0];
  floatx16& u1 = CB == 2 ? acc[1][0] : acc[0][NB - 1];
  floatx16 own;  // the finished unit of this wave
  auto reduce_scatter = [&]() {
    float* red = xs;
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wn * 16 + r) * 64 + lane] = u1[r];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((4 + wn) * 16 + r) * 64 + lane] = u0[r];
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) own[r] = u0[r] + red[((4 + wn) * 16 + r) * 64 + lane];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) own[r] = u1[r] + red[(wn * 16 + r) * 64 + lane];
    }
  };
  const int mb_own = CB == 2 ? kg : 0, nb_own = CB == 2 ? 0 : kg;  // the unit this wave finishes

  // ---- phase 1: conv1 (dilation d) on T1 columns
  zero_acc();
  pair_mfma_phase<K, CB, NB, NO>(acc, wq1, kg, xs + half * XW + shift + wn * (NB * 32) + col, XW, a.dil);
  reduce_scatter();
  {
    // park lrelu(conv1 + bias); columns outside the sequence are conv2's ZERO padding
    float bb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = a.b1[mb_own * 32 + (r & 3) + 8 * (r >> 2) + rbase];
    const int jj = (wn * NB + nb_own) * 32 + col;
    const int g = gt0 + jj;
    const bool inside = g >= 0 && g < L;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mb_own * 32 + (r & 3) + 8 * (r >> 2) + rbase;
      float v = own[r] + bb[r];
      v = v > 0.f ? v : v * slope;
      ts[row * TW + jj] = inside ? v : 0.f;
    }
  }
  __syncthreads();

  // ---- phase 2: conv2 (dilation 1) on the parked tile
  zero_acc();
  pair_mfma_phase<K, CB, NB, NO>(acc, wq2, kg, ts + half * TW + wn * (NB * 32) + col, TW, 1);
  reduce_scatter();
  // ---- epilogue: + bias + residual, batched loads from clamped addresses
  {
    float bb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[r] = a.b2[mb_own * 32 + (r & 3) + 8 * (r >> 2) + rbase];
    const int jj = (wn * NB + nb_own) * 32 + col;
    const int g = j0 + jj;
    const bool tok = jj < T2 && g < L;
    const int gc = g < L ? g : L - 1;
    const float* rb = xb + gc;
    float* yb = a.y + (long long)b * a.bs + gc;
    float rv[16], v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = rb[(mb_own * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (own[r] + bb[r] + rv[r]) * a.alpha;
    if (a.accum) {
      float ov[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ov[r] = yb[(mb_own * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] += ov[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (tok) yb[(mb_own * 32 + (r & 3) + 8 * (r >> 2) + rbase) * a.ld] = v[r];
  }
}

template <int K, int CB, int NB>
__global__ __launch_bounds__(512, CB == 2 ? 4 : 1) void resblock_pair_kernel(const PairArgs a) {
  __shared__ float xs[PairGeom<K, CB, NB>::LDS];
  float* const ts = xs + PairGeom<K, CB, NB>::RED;
  int tile_x, tile_y;
  int gx = gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (conv_mfma.h, row_tiles)
    gx = row_tiles(a.len ? a.len[blockIdx.z] * a.len_mul : a.len_const, PairGeom<K, CB, NB>::T1 - (K - 1));
    if ((int)blockIdx.x >= gx) return;
  }
  xcd_tile_lin(blockIdx.x, gx, 1, tile_x, tile_y);  // neighbouring tiles share their halo through one XCD's L2
  pair_tile<K, CB, NB>(a, tile_x, blockIdx.z, xs, ts);
}

// The fused steps of the three MRF chains of a stage in ONE launch (see conv_group_kernel): the
// first off[1] workgroups are member 0's tiles (largest K first), then member 1's, then member 2's.
struct PairGroupArgs {
  PairArgs p[3];
  int gx[3];
  int off[4];
};
template <int K0, int K1, int K2, int CB, int NB>
// (C = 64: capped at 128 VGPRs so that two of the 52 KB workgroups share a CU)
__global__ __launch_bounds__(512, CB == 2 ? 4 : 1) void pair_group_kernel(const PairGroupArgs g) {
  constexpr int L0 = PairGeom<K0, CB, NB>::LDS, L1 = PairGeom<K1, CB, NB>::LDS, L2 = PairGeom<K2, CB, NB>::LDS;
  __shared__ float xs[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  float* const ts = xs + PairGeom<K0, CB, NB>::RED;
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  // ragged batch: a row deals only its own tiles (conv_mfma.h, row_tiles)
  auto tiles = [&](const PairArgs& p, int gx_grid, int t2) { return gridDim.z > 1 ? row_tiles(p.len ? p.len[b] * p.len_mul : p.len_const, t2) : gx_grid; };
  constexpr int T1 = PairGeom<K0, CB, NB>::T1;
  int tx, ty;
  if (lin < g.off[1]) {
    const int gx = tiles(g.p[0], g.gx[0], T1 - (K0 - 1));
    if (lin >= gx) return;
    xcd_tile_lin(lin, gx, 1, tx, ty);
    pair_tile<K0, CB, NB>(g.p[0], tx, b, xs, ts);
  } else if (lin < g.off[2]) {
    const int l = lin - g.off[1];
    const int gx = tiles(g.p[1], g.gx[1], T1 - (K1 - 1));
    if (l >= gx) return;
    xcd_tile_lin(l, gx, 1, tx, ty);
    pair_tile<K1, CB, NB>(g.p[1], tx, b, xs, ts);
  } else {
    const int l = lin - g.off[2];
    const int gx = tiles(g.p[2], g.gx[2], T1 - (K2 - 1));
    if (l >= gx) return;
    xcd_tile_lin(l, gx, 1, tx, ty);
    pair_tile<K2, CB, NB>(g.p[2], tx, b, xs, ts);
  }
}

}  // namespace mi355tts
