// One dilation step of a ResBlock1 chain in the native fp16 mode as ONE launch (hifi_gan/models.py:91-98):
//     y = x + conv2(lrelu(conv1(lrelu(x), dilation d)), dilation 1)
// conv1's output tile never leaves the CU: its accumulators get bias + leaky ReLU, are rounded to fp16 and written to an LDS tile in
// the octet-row layout — which is already conv2's B operand — and conv2 runs from there.  Against two conv_f16 launches this
// saves a launch (at batch 1 a launch costs more than its MFMA work), the HBM round trip of the intermediate plane, and
// conv2's global -> LDS staging.
//
// Tile: a workgroup owns ALL channels (rows = 32 MB WM >= C) of T = 32 NB WN computed columns.  Both passes compute T columns;
// conv2's column q needs conv1's columns q .. q + K - 1, so T - (K - 1) of conv2's columns are complete: tiles advance by
// TO = T - (K - 1) columns (any column is a whole 16-byte unit — tiles need no alignment), 92 % of both passes useful at
// K = 11, T = 128.  conv2's trailing K - 1 columns read the tile's uninitialised padding and are discarded (column j of
// the B operand reaches column j of the result only).
#pragma once
#include "conv_f16.h"

namespace mi355tts {

struct HPairArgs {
  const uint4* x;  // input planes [B][C / 8][ld] (also the residual)
  uint4* y;        // output planes, same geometry
  long long bs;    // units per batch row
  int ld;          // units per octet row
  const int* len;  // valid columns of row b: len ? len[b] * len_mul : len_const
  int len_mul, len_const;
  const uint4* w1;  // conv1 fragments (pack_conv_f16), dilation `dil`
  const float* b1;
  int nslab1;
  const uint4* w2;  // conv2 fragments, dilation 1
  const float* b2;
  int nslab2;
  int C, dil;
  float slope;  // leaky-ReLU slope of both activations (0.1)
};

constexpr int PAIR_TPAD = 16;  // padding columns of the intermediate tile (>= K - 1)
#ifndef PAIR_F16_ADIST
#define PAIR_F16_ADIST 2  // weight fragments in flight (steps): three or four waves per SIMD cover the rest of an L2 hit between them
#endif

// conv1's staging ring and the intermediate tile SHARE the workgroup's LDS: the ring is dead once every wave has left pass 1's
// main loop (one barrier), and the tile is written only after it — half the LDS of keeping both, twice the workgroups per CU
template <int K, int NB, int WN, int HALO, int CH, int RING>
constexpr int pair_f16_lds_units(int rows) {
  constexpr int ring = conv_f16_lds_units<NB, WN, HALO, CH, RING>();
  const int tile = (rows / 8) * (32 * NB * WN + PAIR_TPAD);
  return ring > tile ? ring : tile;
}

template <int K, int MB, int NB, int WM, int WN, int HALO, int CH, int RING>
__device__ __forceinline__ void pair_f16_tile(const HPairArgs& a, const int tile_x, const int b, uint4* __restrict__ lds) {
  constexpr int T = 32 * NB * WN;
  constexpr int TO = T - (K - 1);
  constexpr int TW = T + PAIR_TPAD;
  constexpr int AD = PAIR_F16_ADIST, BD = F16_BDIST;
  static_assert(K - 1 <= PAIR_TPAD, "intermediate tile padding");
  uint4* const xs = lds;  // conv1's staging ring
  uint4* const ts = lds;  // intermediate tile [rows / 8][TW], over the ring once pass 1's main loop is done

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int mt0 = wm * MB;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int t0 = tile_x * TO;  // first output column of the tile
  if (t0 >= L) return;
  constexpr int P2 = (K - 1) / 2;
  const int c1 = t0 - P2;  // column of conv1's computed column 0

  // ---- pass 1: conv1 over the staged input
  HConvArgs a1;
  a1.x = a.x;
  a1.x2 = nullptr;
  a1.x3 = nullptr;
  a1.in_div = 1.0f;
  a1.x_bs = a.bs;
  a1.x_ld = a.ld;
  a1.w = a.w1;
  a1.bias = a.b1;
  a1.nslab = a.nslab1;
  a1.Cin = a.C;
  a1.dil = a.dil;
  a1.pad = P2 * a.dil;
  a1.in_slope = a.slope;
  floatx16 acc[MB][NB];
  conv_f16_mainloop<K, MB, NB, WM, WN, HALO, CH, false, RING, AD>(a1, c1, mt0, b, L, xs, acc);
  __syncthreads();  // every wave has read its last B fragments: the ring's LDS becomes the intermediate tile

  const int col = lane & 31;
  const int rsub = 4 * (lane >> 5);
  const int rows_t = ((a.C + 31) >> 5) << 5;  // rows of the intermediate tile: whole 32-channel chunks (rows past C are zeros)
  // conv1 epilogue -> intermediate tile: lrelu(acc + bias) inside the sequence, 0 outside (conv2's zero padding)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row0 = (mt0 + mb) * 32 + 8 * j + rsub;
      if (row0 >= rows_t) continue;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int jc = (wn * NB + nb) * 32 + col;
        const int c = c1 + jc;
        const bool in = c >= 0 && c < L;
        half4 hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[mb][nb][4 * j + e];
          v = v > 0.f ? v : v * a.slope;
          hv[e] = (_Float16)(in ? v : 0.f);
        }
        reinterpret_cast<uint2*>(ts + (row0 >> 3) * TW + jc)[(row0 >> 2) & 1] = __builtin_bit_cast(uint2, hv);
      }
    }
  }
  __syncthreads();

  // the residual (the tile's own input columns: L2-warm) is requested HERE, a whole pass ahead of the epilogue that adds it
  const uint4* xb = a.x + (long long)b * a.bs;
  uint4* yb = a.y + (long long)b * a.bs;
  const int hi = lane >> 5;
  uint4 resv[MB][NB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int orow = (mt0 + mb) * 32 + 16 * jp + 8 * hi;
        const int c = t0 + (wn * NB + nb) * 32 + col;
        const bool ok = orow < a.C && c < L;
        const long long off = (long long)(ok ? (orow >> 3) : 0) * a.ld + (ok ? c : 0);
        resv[mb][nb][jp] = xb[off];  // clamped address: nothing behind a branch
      }

  // ---- pass 2: conv2 (dilation 1) from the intermediate tile; weights streamed as in pass 1, no staging, no barriers.
  // Step order = conv_f16_mainloop's at CH = 32 (per 32-channel chunk: tap-major, the chunk's two slabs per tap), so the
  // accumulation order — and with it every bit of the result — is that of the two-launch form.
  const int nch2 = rows_t >> 5;
  constexpr int S2 = 2 * K;
  const uint4* wq[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) wq[mb] = a.w2 + (long long)(mt0 + mb) * a.nslab2 * K * 64;  // wave-uniform bases
  constexpr int CHW2 = 2 * K * 64;
  static_assert(AD <= 2 * S2, "weight look-ahead spans at most two chunk seams");
  auto chunk_base = [&](int c) -> int { return (c < nch2 ? c : nch2 - 1) * CHW2; };
  auto a_off = [&](const int (&cb)[3], int t) -> int {  // step t (compile-time) counted from the running chunk: see conv_f16_mainloop
    const int ci = t / S2, tt = t - ci * S2;
    const int k = tt >> 1, sl = tt & 1;
    return cb[ci] + (sl * K + k) * 64;
  };
  // scalar base + the lane's constant 32-bit byte offset: global_load's saddr form, no vector address arithmetic
  const unsigned lane16 = (unsigned)lane * 16u;
  auto aload = [&](const uint4* base, int soff) -> uint4 {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + soff) + lane16);
  };
  const int colb = wn * (NB * 32) + (lane & 31);
  const int ohalf = lane >> 5;
  auto bread = [&](int ch, int st, uint4* bf) {  // step st of chunk ch; past the last chunk: a harmless re-read of the last one
    ch = ch < nch2 ? ch : nch2 - 1;
    const int k = st >> 1, sl = st & 1;
    const uint4* bp = ts + (4 * ch + 2 * sl + ohalf) * TW + colb + k;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bf[nb] = bp[nb * 32];
  };
  static_assert(BD < S2, "pass 2 looks at most one chunk ahead");
  uint4 Af[AD + 1][MB];
  uint4 Bf[BD + 1][NB];
  {
    const int cb[3] = {chunk_base(0), chunk_base(1), chunk_base(2)};
#pragma unroll
    for (int d = 0; d < AD; ++d)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) Af[d][mb] = aload(wq[mb], a_off(cb, d));
  }
#pragma unroll
  for (int d = 0; d < BD; ++d) bread(0, d, Bf[d]);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {  // accumulators start at conv2's bias (see conv_f16_mainloop)
    float4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4*>(a.b2 + (mt0 + mb) * 32 + 8 * j + 4 * (lane >> 5));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[mb][nb][4 * j + 0] = b4[j].x;
        acc[mb][nb][4 * j + 1] = b4[j].y;
        acc[mb][nb][4 * j + 2] = b4[j].z;
        acc[mb][nb][4 * j + 3] = b4[j].w;
      }
  }
  constexpr int NMF = MB * NB;
  for (int ch = 0; ch < nch2; ++ch) {
    const int cb[3] = {chunk_base(ch), chunk_base(ch + 1), chunk_base(ch + 2)};
#pragma unroll
    for (int st = 0; st < S2; ++st) {
      {
        const int off = a_off(cb, st + AD);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) Af[AD][mb] = aload(wq[mb], off);
      }
      if (st + BD < S2) bread(ch, st + BD, Bf[BD]);
      else bread(ch + 1, st + BD - S2, Bf[BD]);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma_f16(Af[0][mb], Bf[0][nb], acc[mb][nb]);
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < MB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        else if (i - MB < NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (NMF - MB < NB) __builtin_amdgcn_sched_group_barrier(0x100, NB - (NMF - MB), 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < AD; ++d)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) Af[d][mb] = Af[d + 1][mb];
#pragma unroll
      for (int d = 0; d < BD; ++d)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) Bf[d][nb] = Bf[d + 1][nb];
    }
  }

  // ---- conv2 epilogue: y = acc + bias + x (residual in f32, one rounding); whole 16-byte units per lane (see conv_f16_tile)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int q = (wn * NB + nb) * 32 + col;
      const int c = t0 + q;
      if (q >= TO || c >= L) continue;  // (lanes l and l + 32 share their column)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int orow = (mt0 + mb) * 32 + 16 * jp + 8 * hi;
        const bool ok = orow < a.C;
        const long long off = (long long)(orow >> 3) * a.ld + c;
        const uint4 rv = resv[mb][nb][jp];
        uint2 ra = uint2{rv.x, rv.y}, rb = uint2{rv.z, rv.w};
        swap32(ra, rb);
        uint2 u[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          const half4 rh = __builtin_bit_cast(half4, jj ? rb : ra);
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(acc[mb][nb][4 * j + e] + (float)rh[e]);
          u[jj] = __builtin_bit_cast(uint2, hv);
        }
        swap32(u[0], u[1]);
        if (ok) yb[off] = uint4{u[0].x, u[0].y, u[1].x, u[1].y};
      }
    }
  }
}

// tiles of a row of n columns
template <int K, int NB, int WN>
__device__ __forceinline__ int pair_f16_tiles(int n) {
  constexpr int TO = 32 * NB * WN - (K - 1);
  return (n + TO - 1) / TO;
}

// The three MRF chains' same-geometry steps in ONE launch (members longest first; group sizes padded to multiples of 8).
// ROWS = 32 MB WM, the rows a workgroup owns (>= C).
struct HPairGroupArgs {
  HPairArgs p[3];
  int gx[3];   // tiles of the longest row, per member
  int off[4];  // first workgroup of each member, off[3] = grid size
  int n;       // members (1 .. 3)
};
template <int K0, int K1, int K2, int MB, int NB, int WM, int WN, int H0, int H1, int H2, int CH, int RING, int MINW>
__global__ __launch_bounds__(64 * WM * WN, MINW) void pair_f16_group_kernel(const HPairGroupArgs g) {
  constexpr int ROWS = 32 * MB * WM;
  constexpr int L0 = pair_f16_lds_units<K0, NB, WN, H0, CH, RING>(ROWS), L1 = pair_f16_lds_units<K1, NB, WN, H1, CH, RING>(ROWS),
                L2 = pair_f16_lds_units<K2, NB, WN, H2, CH, RING>(ROWS);
  __shared__ uint4 lds[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  const bool ragged = gridDim.z > 1;
  if (lin < g.off[1]) {
    const HPairArgs& a = g.p[0];
    const int gx = ragged ? pair_f16_tiles<K0, NB, WN>(a.len ? a.len[b] * a.len_mul : a.len_const) : g.gx[0];
    if (lin >= gx) return;
    pair_f16_tile<K0, MB, NB, WM, WN, H0, CH, RING>(a, lin, b, lds);
  } else if (lin < g.off[2]) {
    const HPairArgs& a = g.p[1];
    const int l = lin - g.off[1];
    const int gx = ragged ? pair_f16_tiles<K1, NB, WN>(a.len ? a.len[b] * a.len_mul : a.len_const) : g.gx[1];
    if (l >= gx) return;
    pair_f16_tile<K1, MB, NB, WM, WN, H1, CH, RING>(a, l, b, lds);
  } else {
    const HPairArgs& a = g.p[2];
    const int l = lin - g.off[2];
    const int gx = ragged ? pair_f16_tiles<K2, NB, WN>(a.len ? a.len[b] * a.len_mul : a.len_const) : g.gx[2];
    if (l >= gx) return;
    pair_f16_tile<K2, MB, NB, WM, WN, H2, CH, RING>(a, l, b, lds);
  }
}

}  // namespace mi355tts
