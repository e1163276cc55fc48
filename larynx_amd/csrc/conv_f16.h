// The NATIVE 16-bit vocoder — what the reference's `half` switch is: `.half()` on the whole HiFi-GAN generator
// (larynx/hifi_gan.py:96-97; hifi_gan/models.py:91-98, 136-141, 186-202).  fp16 activation planes in HBM between the layers,
// fp16 weights, ONE v_mfma_f32_32x32x16_f16 per product, f32 accumulation; bias, residual and the output activation are
// applied to the f32 accumulator before its ONE rounding to fp16.  (The split-bf16 mode of conv_bf16.h keeps f32 planes and
// pays three MFMAs per product for f32-class accuracy; this one trades accuracy for speed, as the reference's does.)
//
// Layout of an activation plane: "octet rows" — [C / 8][ld][8] halves: the 8 channels 8o .. 8o + 7 of one time column are one
// 16-byte unit, units of an octet run along time.  That is exactly the B operand of the MFMA (lane l: column l & 31, channels
// 8 (l >> 5) .. + 7 of a 16-channel slab), so
//   * staging is a COPY of 16-byte units global -> LDS (input leaky-ReLU as packed-half max(v, v * slope) on the way; no
//     transposition, no conversion), one B fragment = ONE aligned conflict-free ds_read_b128 at any tap offset,
//   * the C/D layout of the 32 x 32 MFMA gives a lane 4 consecutive channels of one column: one 8-byte store per 8-row group,
//     the two lane halves completing each 16-byte unit.
// GEMM view as in conv_mfma.h: M = output channels (A = weights pre-packed per (m-tile, 16-channel slab, tap) as one 16-byte
// load per lane, streamed L2 -> VGPR two steps ahead), N = time, K-dim = (slab, tap).  A wave owns MB x NB blocks of 32 x 32
// over ALL input channels (no k-split); a workgroup = WM x WN waves share the staged [CH channels x (tile + halo)] chunk.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_mfma.h"

namespace mi355tts {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

struct HConvArgs {
  // input planes [B][ceil(Cin / 8)][x_ld] units (16 bytes = 8 halves); optional x2 / x3 of the same geometry are averaged in on load:
  // ((x + x2) + x3) / in_div in f32, one rounding (the MRF average of hifi_gan/models.py:191-197, taken by the consumer)
  const uint4* x;
  const uint4* x2;
  const uint4* x3;
  float in_div;
  long long x_bs;  // units per batch row
  int x_ld;        // units per octet row
  const int* in_len;  // valid input columns of row b: in_len ? in_len[b] * in_mul : in_const
  int in_mul, in_const;
  const uint4* w;     // [m-tile][slab][tap][64 lanes] fragments (pack_conv_f16)
  const float* bias;  // [rows] f32, virtual-row order
  int nslab;          // 16-channel slabs in the packed weights (whole staged chunks)
  int Cin, rows, dil, pad;
  float in_slope;   // leaky-ReLU slope applied to x on load (1 = identity)
  float out_slope;  // leaky-ReLU slope applied to the result before its rounding (1 = identity): conv1 of a ResBlock1 step
                    // stores lrelu(conv1(.)) — its only consumer is conv2, which would apply it on load
  uint4* y;         // output planes [B][rows / 8][y_ld] units (EPI_UPSAMPLE: [B][cout / 8][y_ld])
  long long y_bs;
  int y_ld;
  const uint4* res;  // optional residual, geometry of y (added in f32 before the rounding)
  const int* out_len;
  int out_mul, out_const;
  // EPI_UPSAMPLE (polyphase ConvTranspose1d): virtual row v = r * cout + co (phase-major: a lane's 4 consecutive rows are 4
  // consecutive CHANNELS of one output sample) goes to y[co][q * up + r - up_pad]
  int up, up_pad, cout;
  int rows_major;
};

// staged halo columns per tap count: (K - 1) x the largest dilation the reference's vocoder configs use with that tap count
template <int K> struct ConvHalo;
template <> struct ConvHalo<1> { static constexpr int v = 0; };
template <> struct ConvHalo<2> { static constexpr int v = 4; };
template <> struct ConvHalo<3> { static constexpr int v = 12; };
template <> struct ConvHalo<5> { static constexpr int v = 24; };
template <> struct ConvHalo<7> { static constexpr int v = 72; };
template <> struct ConvHalo<11> { static constexpr int v = 52; };

__device__ __forceinline__ floatx16 mfma_f16(const uint4& a, const uint4& b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// lanes 32 .. 63 of `a` change places with lanes 0 .. 31 of `b` (v_permlane32_swap_b32), both dwords
__device__ __forceinline__ void swap32(uint2& a, uint2& b) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
  a = uint2{r0[0], r1[0]};
  b = uint2{r0[1], r1[1]};
}

// leaky ReLU of 8 packed halves, 0 <= slope <= 1: max(v, v * slope)
__device__ __forceinline__ uint4 lrelu_h8(const uint4& v, _Float16 slope) {
  const half8 h = __builtin_bit_cast(half8, v);
  const half8 m = h * slope;
  return __builtin_bit_cast(uint4, __builtin_elementwise_max(h, m));
}

#ifndef F16_ADIST
#define F16_ADIST 4  // weight fragments in flight ahead of the MFMAs that use them (steps)
#endif
#ifndef F16_ABLATE
#define F16_ABLATE 0  // probe builds only (tools/probe/f16_bench.hip): 1 = no epilogue, 2 = no MFMAs, 4 = no weight loads, 8 = no LDS reads, 16 = no staging
#endif
#ifndef F16_STAMPS  // probe builds only: thread 0 of every workgroup writes the cycle counter at phase boundaries to g_f16_stamps[wg][8]
#define F16_STAMPS 0
#endif
#if F16_STAMPS
__device__ unsigned long long* g_f16_stamps = nullptr;
#define F16_STAMP(i)                                                                                                                \
  do {                                                                                                                              \
    if (threadIdx.x == 0 && g_f16_stamps) g_f16_stamps[(size_t)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define F16_STAMP(i) ((void)0)
#endif
#ifndef F16_MIN_WAVES
#define F16_MIN_WAVES 1  // __launch_bounds__' second argument: waves per SIMD the register allocation must leave room for
#endif
#ifndef F16_BDIST
#define F16_BDIST 2  // LDS fragments ahead (steps)
#endif

// RING = staged chunks the workgroup keeps in LDS: 3 = the ring (any number of chunks); 1 / 2 = a conv of at most that many chunks,
// all staged in the prologue (a third or two thirds of the ring's LDS: more workgroups per CU on the narrow stages)
template <int NB, int WN, int HALO, int CH, int RING = 3>
constexpr int conv_f16_lds_units() {
  return RING * (CH / 8) * (32 * NB * WN + HALO);
}

// The main loop of a tile: acc[mb][nb] = bias + W (m-tiles mt0 .. mt0 + MB) x X over all input channels and taps, for the T_T
// computed columns whose first one is implicit-GEMM column t0 (input column t0 - a.pad at tap 0).  `a.w`, `a.bias`, `a.nslab`,
// `a.Cin`, `a.dil`, `a.pad`, `a.in_slope`, the input planes and `Lin` (valid input columns) are what it reads of the arguments.
// The accumulators START at the bias (virtual-row order, padded to whole m-tiles by pack_conv_f16): its loads — four 16-byte
// loads per m-tile — fly with the prologue's staging loads instead of standing between the last MFMA and the stores.
template <int K, int MB, int NB, int WM, int WN, int HALO, int CH, bool MRF, int RING = 3, int AD = F16_ADIST>
__device__ __forceinline__ void conv_f16_mainloop(const HConvArgs& a, const int t0, const int mt0, const int b, const int Lin, uint4* __restrict__ xs,
                                                  floatx16 (&acc)[MB][NB]) {
  static_assert(CH == 32 || CH == 64, "staged chunk: 32 or 64 channels");
  constexpr int NT = 64 * WM * WN;
  constexpr int T_T = 32 * NB * WN;  // time columns per workgroup
  constexpr int XW = T_T + HALO;     // staged columns per octet row
  constexpr int OCT = CH / 8;
  constexpr int BUF = OCT * XW;      // uint4 units per buffer
  constexpr int NU = (BUF + NT - 1) / NT;
  constexpr int SPC = CH / 16;       // slabs per chunk
  constexpr int S = SPC * K;         // steps per chunk: tap-major, the chunk's slabs per tap
  constexpr int BD = F16_BDIST;
  static_assert(BD >= 1 && BD <= S && AD >= 1, "pipeline depths");
  static_assert(RING >= 1 && RING <= 3, "ring depth");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WM;

  const uint4* xb = a.x + (long long)b * a.x_bs;
  const uint4* xb2 = MRF ? a.x2 + (long long)b * a.x_bs : nullptr;
  const uint4* xb3 = (MRF && a.x3) ? a.x3 + (long long)b * a.x_bs : nullptr;
  const int noct_in = (a.Cin + 7) >> 3;
  const int nchunks = (a.Cin + CH - 1) / CH;
  const int c_first = t0 - a.pad;  // input column of staged column 0
  const int ld_last = a.x_ld - 1;
  const _Float16 slope = (_Float16)a.in_slope;
  const float inv_div = 1.0f / (MRF ? a.in_div : 1.0f);
  const bool plain = a.in_slope == 1.0f;

  // ---- staging: unit u = (octet o, staged column c): one 16-byte load (clamped address: nothing behind a branch), masked,
  // activated, one ds_write_b128.  THREE buffers: chunk c + 2 is requested at the start of chunk c's MFMA phase and written at
  // its end, in front of the ONE barrier of the chunk — which publishes it a whole chunk before its first read, so the B
  // fragments of the next chunk's first steps are read across the chunk boundary and the matrix stream never drains at a seam
  // (rb_conv.h's ring, for this tile).  Chunks past the last one are not staged (wave-uniform guards: a 32-channel conv has ONE
  // chunk, and staging two more that nobody reads tripled its load and VALU work).
  auto gload = [&](int chunk, uint4 (&pre)[MRF ? 3 : 1][NU]) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = tid + NT * i;
      const int uu = u < BUF ? u : BUF - 1;
      const int o = uu / XW, c = uu - o * XW;
      const int col = c_first + c;
      const int cc = col < 0 ? 0 : (col > ld_last ? ld_last : col);
      int oct = chunk * OCT + o;
      oct = oct < noct_in ? oct : noct_in - 1;
      const long long off = (long long)oct * a.x_ld + cc;
      pre[0][i] = xb[off];
      if constexpr (MRF) {
        pre[1][i] = xb2[off];
        pre[2][i] = xb3 ? xb3[off] : uint4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto lstore = [&](int buf, int chunk, const uint4 (&pre)[MRF ? 3 : 1][NU]) {
#pragma unroll
    for (int i = 0; i < NU; ++i) {
      const int u = tid + NT * i;
      if (BUF % NT != 0 && u >= BUF) continue;
      const int o = u / XW, c = u - o * XW;
      const int col = c_first + c;
      const bool ok = col >= 0 && col < Lin && chunk * OCT + o < noct_in;
      uint4 v = pre[0][i];
      if constexpr (MRF) {
        // the MRF average (xs / num_kernels, hifi_gan/models.py:191-197) summed in f32 and rounded ONCE — v_fma_mix takes the
        // half operands as they are, and the division is a multiplication by 1 / num_kernels.  (Packed-half sums, which is how
        // the reference's half tensors take it, cost a fifth of the first form's sixty VALU operations per unit but put the
        // 'high' goldens at 3.36e-4 RMS against the reference's own 3.30e-4: this form keeps them at 2.7e-4.)
        const half8 h0 = __builtin_bit_cast(half8, v), h1 = __builtin_bit_cast(half8, pre[1][i]), h2 = __builtin_bit_cast(half8, pre[2][i]);
        half8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (_Float16)((((float)h0[e] + (float)h1[e]) + (float)h2[e]) * inv_div);
        v = __builtin_bit_cast(uint4, r);
      }
      if (!plain) v = lrelu_h8(v, slope);
      if (!ok) v = uint4{0u, 0u, 0u, 0u};
      xs[buf * BUF + u] = v;
    }
  };

#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    float4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = *reinterpret_cast<const float4*>(a.bias + (mt0 + mb) * 32 + 8 * j + 4 * (lane >> 5));
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
  uint4 pre[MRF ? 3 : 1][NU];

  // A stream of m-tile mt: uint4 index ((mt * nslab + slab) * K + k) * 64 + lane.  The base of an m-tile is wave-uniform (a
  // scalar pointer), the lane's share a constant 32-bit offset, and the step's offset = its chunk's base + a COMPILE-TIME
  // constant (the steps of a chunk are unrolled): no division, no 64-bit vector adds in the loop (the first form derived the
  // offset from the global step number, 21 scalar instructions a step in front of the step's first MFMA).  Steps past the
  // last chunk re-read the last chunk (chunk index clamped): loaded, never used.
  const uint4* wq[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) wq[mb] = a.w + (long long)(mt0 + mb) * a.nslab * K * 64;
  constexpr int CHW = SPC * K * 64;  // uint4 units of one chunk's fragments of one m-tile
  static_assert(AD <= 2 * S, "weight look-ahead spans at most two chunk seams");
  auto chunk_base = [&](int c) -> int { return (c < nchunks ? c : nchunks - 1) * CHW; };
  // offset of step t (0 <= t < 3 S, compile-time after unrolling) counted from the running chunk, whose and whose two
  // successors' bases are cb[0 .. 2]
  auto a_off = [&](const int (&cb)[3], int t) -> int {
    const int ci = t / S, tt = t - ci * S;
    const int k = tt / SPC, sl = tt - k * SPC;
    return cb[ci] + (sl * K + k) * 64;
  };
  // scalar base + the lane's constant 32-bit byte offset: global_load's saddr form, no vector address arithmetic
  const unsigned lane16 = (unsigned)lane * 16u;
  auto aload = [&](const uint4* base, int soff) -> uint4 {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + soff) + lane16);
  };
  // Operand pipeline of a wave (registers): weight fragments AD steps ahead of the MFMAs that use them (an L2 hit is ~700
  // cycles, a step of MB x NB MFMAs 128-256: with two steps ahead a lone wave per SIMD ran at 0.2 of its matrix pipe), LDS
  // fragments BD steps ahead; every load is pinned behind one MFMA with sched_group_barrier (left alone the compiler sinks them
  // to their first use).
  uint4 Af[AD + 1][MB];
  uint4 Bf[BD + 1][NB];

  const int colb = wn * (NB * 32) + (lane & 31);
  const int ohalf = lane >> 5;
  auto bread = [&](int buf, int st, uint4* bf) {  // step st of the chunk staged in ring buffer `buf`
    const int k = st / SPC, s = st - k * SPC;
    const uint4* bp = xs + buf * BUF + colb + (2 * s + ohalf) * XW + k * a.dil;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bf[nb] = bp[nb * 32];
  };

  gload(0, pre);
  {
    const int cb[3] = {chunk_base(0), chunk_base(1), chunk_base(2)};
#pragma unroll
    for (int d = 0; d < AD; ++d)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) Af[d][mb] = aload(wq[mb], a_off(cb, d));
  }
  lstore(0, 0, pre);
  if (RING > 1 && nchunks > 1) {
    gload(1, pre);
    lstore(1, 1, pre);
  }
  __syncthreads();
  F16_STAMP(1);
#pragma unroll
  for (int d = 0; d < BD; ++d) bread(0, d, Bf[d]);

  constexpr int NMF = MB * NB;  // MFMAs per step
  int buf = 0;                  // ring buffer of the running chunk
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf1 = buf == RING - 1 ? 0 : buf + 1, buf2 = RING < 3 ? 0 : (buf1 == 2 ? 0 : buf1 + 1);
    const bool more = RING == 3 && chunk + 2 < nchunks && !(F16_ABLATE & 16);
    if (chunk == 2) F16_STAMP(5);
    if (more) gload(chunk + 2, pre);
    const int cb[3] = {chunk_base(chunk), chunk_base(chunk + 1), chunk_base(chunk + 2)};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < S; ++st) {
      {
        const int off = a_off(cb, st + AD);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          if (!(F16_ABLATE & 4)) Af[AD][mb] = aload(wq[mb], off);
      }
      if (!(F16_ABLATE & 8)) {
        if (st + BD < S) bread(buf, st + BD, Bf[BD]);
        else bread(buf1, st + BD - S, Bf[BD]);  // across the seam: the next chunk's tile was published a chunk ago
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          if (!(F16_ABLATE & 2)) acc[mb][nb] = mfma_f16(Af[0][mb], Bf[0][nb], acc[mb][nb]);
      // issue order: the MB weight loads behind the first MFMAs, then the NB LDS reads (0x008 = MFMA, 0x020 = VMEM read, 0x100 = LDS read)
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
    if (more) lstore(buf2, chunk + 2, pre);
    if (chunk == 2) F16_STAMP(6);
    if (chunk + 1 < nchunks) __syncthreads();
    if (chunk == 2) F16_STAMP(7);
    buf = buf1;
  }
  F16_STAMP(2);
}

template <int K, int MB, int NB, int WM, int WN, int HALO, int CH, int EPI, bool MRF, int RING = 3>
__device__ __forceinline__ void conv_f16_tile(const HConvArgs& a, const int tile_x, const int tile_y, const int b, uint4* __restrict__ xs) {
  static_assert(EPI == EPI_LINEAR || EPI == EPI_UPSAMPLE, "f16 tile epilogues");
  constexpr int T_T = 32 * NB * WN;  // time columns per workgroup
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int t0 = tile_x * T_T;
  const int mt0 = (tile_y * WM + wm) * MB;

  const int Lin = a.in_len ? a.in_len[b] * a.in_mul : a.in_const;
  const int Lout = a.out_len ? a.out_len[b] * a.out_mul : a.out_const;
  const int n_len = (EPI == EPI_UPSAMPLE) ? (Lin > 0 ? Lin + (K - 1) : 0) : Lout;
  if (t0 >= n_len) return;  // uniform per workgroup
  F16_STAMP(0);

  // the residual is requested BEFORE the main loop (clamped addresses: nothing behind a branch) and consumed after it
  const int col = lane & 31;
  const int hi = lane >> 5;
  uint4 resv[MB][NB][2];
  if (EPI == EPI_LINEAR && a.res) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const int orow = (mt0 + mb) * 32 + 16 * jp + 8 * hi;
          const int q = t0 + (wn * NB + nb) * 32 + col;
          const bool ok = orow < a.rows && q < n_len;
          resv[mb][nb][jp] = a.res[(long long)b * a.y_bs + (long long)(ok ? (orow >> 3) : 0) * a.y_ld + (ok ? q : 0)];
        }
  }
  floatx16 acc[MB][NB];
  conv_f16_mainloop<K, MB, NB, WM, WN, HALO, CH, MRF, RING>(a, t0, mt0, b, Lin, xs, acc);
  if (F16_ABLATE & 1) {  // probe builds: keep the accumulators alive, store nothing
    float t = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) t += acc[mb][nb][0] + acc[mb][nb][7];
    if (t == 1.2345e30f) a.y[0] = uint4{0u, 0u, 0u, 0u};
    return;
  }

  // ---- epilogue: a lane holds, per 32 x 32 block, column l & 31 and the rows 8 j + 4 (l >> 5) + (0 .. 3), j = 0 .. 3: HALF an
  // octet unit per j.  Lanes l and l + 32 trade halves (swap32): of the octets j0 = 2 jp and j1 = 2 jp + 1 the lower lane ends up
  // with all of j0, the upper lane with all of j1 — residual loads and result stores are whole 16-byte units, half as many
  // memory instructions as 8 bytes per lane (the epilogue's cost is their issue, not their bytes).
  const bool out_act = a.out_slope != 1.0f;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int q = t0 + (wn * NB + nb) * 32 + col;
      if (q >= n_len) continue;  // (lanes l and l + 32 share their column: both stay or both go)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp) {
        const int orow = (mt0 + mb) * 32 + 16 * jp + 8 * hi;  // first row of the octet this lane loads / stores
        bool ok = orow < a.rows;
        long long off;
        if constexpr (EPI == EPI_UPSAMPLE) {
          const int r = orow / a.cout, co0 = orow - r * a.cout;
          const int n = q * a.up + r - a.up_pad;
          ok = ok && n >= 0 && n < Lout;
          off = (long long)b * a.y_bs + (long long)(co0 >> 3) * a.y_ld + n;
        } else {
          off = (long long)b * a.y_bs + (long long)(orow >> 3) * a.y_ld + q;
        }
        uint2 ra = {0u, 0u}, rb = {0u, 0u};  // residuals of j0 and j1 in the accumulators' layout
        if (EPI == EPI_LINEAR && a.res) {
          const uint4 rv = resv[mb][nb][jp];
          ra = uint2{rv.x, rv.y};
          rb = uint2{rv.z, rv.w};
          swap32(ra, rb);
        }
        uint2 u[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          const half4 rh = __builtin_bit_cast(half4, jj ? rb : ra);
          half4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[mb][nb][4 * j + e];
            if (EPI == EPI_LINEAR && a.res) v += (float)rh[e];
            if (EPI == EPI_LINEAR && out_act) v = v > 0.f ? v : v * a.out_slope;
            hv[e] = (_Float16)v;
          }
          u[jj] = __builtin_bit_cast(uint2, hv);
        }
        swap32(u[0], u[1]);
        if (ok) a.y[off] = uint4{u[0].x, u[0].y, u[1].x, u[1].y};
      }
    }
  }
  F16_STAMP(3);
#if F16_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  F16_STAMP(4);
#endif
}

template <int K, int EPI>
__device__ __forceinline__ int conv_f16_n_len(const HConvArgs& a, int b) {
  if (EPI == EPI_UPSAMPLE) {
    const int Lin = a.in_len ? a.in_len[b] * a.in_mul : a.in_const;
    return Lin > 0 ? Lin + (K - 1) : 0;
  }
  return a.out_len ? a.out_len[b] * a.out_mul : a.out_const;
}

template <int K, int MB, int NB, int WM, int WN, int HALO, int CH, int EPI, bool MRF, int RING = 3>
__global__ __launch_bounds__(64 * WM * WN, F16_MIN_WAVES) void conv_f16_kernel(const HConvArgs a) {
  __shared__ uint4 xs[conv_f16_lds_units<NB, WN, HALO, CH, RING>()];
  int tile_x, tile_y;
  int gx = gridDim.x;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (conv_mfma.h, row_tiles)
    gx = row_tiles(conv_f16_n_len<K, EPI>(a, blockIdx.z), 32 * NB * WN);
    if (lin >= gx * (int)gridDim.y) return;
  }
  xcd_tile_lin(lin, gx, gridDim.y, tile_x, tile_y, a.rows_major);
  conv_f16_tile<K, MB, NB, WM, WN, HALO, CH, EPI, MRF, RING>(a, tile_x, tile_y, blockIdx.z, xs);
}

// The three MRF chains' same-geometry convs in ONE launch (conv_group_kernel's layout: members longest first, group sizes
// padded to multiples of 8 so a tile's XCD stays the one xcd_tile_lin assumes).
struct HConvGroupArgs {
  HConvArgs c[3];
  int gx[3], gy[3];
  int off[4];
};
template <int K0, int K1, int K2, int MB, int NB, int WM, int WN, int H0, int H1, int H2, int CH, int RING = 3>
__global__ __launch_bounds__(64 * WM * WN, F16_MIN_WAVES) void conv_f16_group_kernel(const HConvGroupArgs g) {
  constexpr int L0 = conv_f16_lds_units<NB, WN, H0, CH, RING>(), L1 = conv_f16_lds_units<NB, WN, H1, CH, RING>(), L2 = conv_f16_lds_units<NB, WN, H2, CH, RING>();
  __shared__ uint4 xs[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  const bool ragged = gridDim.z > 1;
  constexpr int T_T = 32 * NB * WN;
  int tx, ty;
  if (lin < g.off[1]) {
    const int gx = ragged ? row_tiles(conv_f16_n_len<K0, EPI_LINEAR>(g.c[0], b), T_T) : g.gx[0];
    if (lin >= gx * g.gy[0]) return;
    xcd_tile_lin(lin, gx, g.gy[0], tx, ty);
    conv_f16_tile<K0, MB, NB, WM, WN, H0, CH, EPI_LINEAR, false, RING>(g.c[0], tx, ty, b, xs);
  } else if (lin < g.off[2]) {
    const int l = lin - g.off[1];
    const int gx = ragged ? row_tiles(conv_f16_n_len<K1, EPI_LINEAR>(g.c[1], b), T_T) : g.gx[1];
    if (l >= gx * g.gy[1]) return;
    xcd_tile_lin(l, gx, g.gy[1], tx, ty);
    conv_f16_tile<K1, MB, NB, WM, WN, H1, CH, EPI_LINEAR, false, RING>(g.c[1], tx, ty, b, xs);
  } else {
    const int l = lin - g.off[2];
    const int gx = ragged ? row_tiles(conv_f16_n_len<K2, EPI_LINEAR>(g.c[2], b), T_T) : g.gx[2];
    if (l >= gx * g.gy[2]) return;
    xcd_tile_lin(l, gx, g.gy[2], tx, ty);
    conv_f16_tile<K2, MB, NB, WM, WN, H2, CH, EPI_LINEAR, false, RING>(g.c[2], tx, ty, b, xs);
  }
}

// ---- f32 [B][C][ld] rows (the mel the acoustic model hands over) -> octet planes [B][ceil(C / 8)][o_ld] (zero channels past C)
__global__ __launch_bounds__(256) void pack_octets_kernel(const float* x, long long x_bs, int x_ld, int C, const int* len, int len_mul,
                                                          uint4* y, long long y_bs, int y_ld) {
  const int b = blockIdx.z, o = blockIdx.y;
  const int L = len[b] * len_mul;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  half8 h;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * o + e;
    h[e] = (_Float16)(c < C ? x[(long long)b * x_bs + (long long)c * x_ld + t] : 0.f);
  }
  y[(long long)b * y_bs + (long long)o * y_ld + t] = __builtin_bit_cast(uint4, h);
}

// ---- conv_post on octet planes: x = tanh(conv_post(leaky_relu(avg(x, x2, x3), slope))) (hifi_gan/models.py:198-201), K taps,
// C <= 64 input channels, ONE output row in f32 + the tile's |max| (voc_out.h's post_conv_kernel, for the f16 planes).
struct HPostArgs {
  const uint4* x;
  const uint4* x2;
  const uint4* x3;
  float in_div, slope;
  long long x_bs;
  int x_ld;
  const int* len;  // row length: len ? len[b] * len_mul : len_const
  int len_mul, len_const;
  const float* w;  // [C][K] f32
  const float* bias;
  int C;
  float* y;  // [B][y_bs] f32
  long long y_bs;
  float* peak;  // optional [B][peak_ld]: |max| of every 256-column tile
  long long peak_ld;
};
constexpr int HPOST_TW = 256;
template <int K, int NPL>
__global__ __launch_bounds__(256) void post_f16_kernel(const HPostArgs a) {
  constexpr int XW = HPOST_TW + K - 1;
  constexpr int MAXOCT = 8;
  __shared__ float xs[MAXOCT * 8][XW + 1];  // [channel][staged column] f32, averaged and activated
  __shared__ float wsm[MAXOCT * 8 * K];
  __shared__ float pm[4];
  const int b = blockIdx.y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int t0 = blockIdx.x * HPOST_TW;
  if (t0 >= L) return;
  const int tid = threadIdx.x;
  const int noct = (a.C + 7) >> 3;
  for (int i = tid; i < a.C * K; i += 256) wsm[i] = a.w[i];
  const uint4* xb = a.x + (long long)b * a.x_bs;
  const uint4* xb2 = NPL > 1 ? a.x2 + (long long)b * a.x_bs : nullptr;
  const uint4* xb3 = NPL > 2 ? a.x3 + (long long)b * a.x_bs : nullptr;
  const float inv_div = 1.0f / a.in_div;
  for (int u = tid; u < noct * XW; u += 256) {
    const int o = u / XW, c = u - o * XW;
    const int col = t0 - (K - 1) / 2 + c;
    const bool ok = col >= 0 && col < L;
    const long long off = (long long)o * a.x_ld + (ok ? col : 0);
    const half8 h0 = __builtin_bit_cast(half8, xb[off]);
    half8 h1 = h0, h2 = h0;
    if (NPL > 1) h1 = __builtin_bit_cast(half8, xb2[off]);
    if (NPL > 2) h2 = __builtin_bit_cast(half8, xb3[off]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = (float)h0[e];
      if (NPL > 1) v += (float)h1[e];
      if (NPL > 2) v += (float)h2[e];
      if (NPL > 1) v = v * inv_div;
      v = v > 0.f ? v : v * a.slope;
      xs[8 * o + e][c] = ok ? v : 0.f;
    }
  }
  __syncthreads();
  const int t = t0 + tid;
  float acc = a.bias ? a.bias[0] : 0.f;
  for (int c = 0; c < a.C; ++c) {
#pragma unroll
    for (int k = 0; k < K; ++k) acc = fmaf(wsm[c * K + k], xs[c][tid + k], acc);
  }
  float m = 0.f;
  if (t < L) {
    const float v = tanhf(acc);
    a.y[(long long)b * a.y_bs + t] = v;
    m = fabsf(v);
  }
  if (a.peak) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((tid & 63) == 0) pm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) a.peak[(long long)b * a.peak_ld + blockIdx.x] = fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3]));
  }
}

}  // namespace mi355tts
