// conv1d / conv-transpose1d / 1x1 as an implicit GEMM on the CDNA4 f32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact f32, bit-equal to an fmaf chain).
//
//   y[co][t] = epi( bias[co] + sum_{ci,k} W[co][ci][k] * act(x[ci][t + k*dil - pad]) )
//
// GEMM view: M = output channels (A = weights, pre-packed in MFMA fragment order
// and streamed straight from L2 into VGPRs, 16 B per lane per load),
// N = time (B = activations, staged through LDS with the input activation applied
// once per element, halo included), K-dim = (ci, tap).
//
// One workgroup = WN x KS waves = (MB*32) output rows x (WN*NB*32) time columns;
// each wave owns MB x NB MFMA blocks of 32x32 (16 accumulator VGPRs each) over its
// k-group's share of the input channels.
//
// This one kernel covers every dense contraction on the Larynx hot path
// (reference ops): HiFi-GAN conv_pre / ResBlock convs / conv_post
// (hifi_gan/models.py:91-98,136-141,186-200), the transposed-conv upsampler in
// polyphase form (:189-190), and the GlowTTS prenet / FFN / 1x1 / WaveNet convs
// (glow_tts/layers.py:73-80,138-162; attentions.py:119-142,375-383).
#pragma once
#include <hip/hip_runtime.h>
#include "prio.h"

// Micro-benchmark ablations (tools/conv_probe.py) are compiled in only for probe builds
// (-DMI355TTS_ABLATION): a runtime test around the weight loads puts them behind a branch,
// and the compiler's s_waitcnt insertion must then assume either path at every join — it
// emitted vmcnt(0) drains inside the MFMA loop, which cost the product kernel ~10 %.
#ifdef MI355TTS_ABLATION
#define MI355TTS_ABLATE(a, bit) ((a).ablate & (bit))
#else
#define MI355TTS_ABLATE(a, bit) 0
#endif

#ifndef CONV_STAMP
#define CONV_STAMP(n)  // phase stamps of tools/probe/glow_conv_bench.hip
#endif
#ifndef CONV_WG_STAMP
#define CONV_WG_STAMP(lin, which)  // workgroup begin / end stamps of tools/probe/rb_diag.hip
#endif
#ifndef CONV_CHUNK_STAMP
#define CONV_CHUNK_STAMP(chunk, which)  // per-chunk phase stamps of tools/probe/rb_diag.hip
#endif
#ifndef MI355TTS_ARING
#define MI355TTS_ARING 3  // depth of the A-fragment register ring (steps in flight + 1)
#endif

namespace mi355tts {


typedef float floatx16 __attribute__((ext_vector_type(16)));

enum ConvEpilogue {
  EPI_LINEAR = 0,    // y = [y +] alpha * (acc + bias [+ res]) ; optional row split into (y, y2)
  EPI_GATE = 1,      // y[c] = tanh(acc[row i]) * sigmoid(acc[row i+16]) (glow_tts/utils.py:31-38)
  EPI_COUPLING = 2,  // y[c] = (res[c] - acc_m) * exp(-acc_logs)        (attentions.py:135-136)
  EPI_UPSAMPLE = 3,  // polyphase ConvTranspose1d scatter: row = co*u + r -> y[co][q*u + r - p]
};

enum OutAct { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct ConvArgs {
  // input activations [B][Cin][x_ld]; optional x2/x3 of the same geometry are
  // summed in on load and the sum divided by in_div (the MRF average
  // xs / num_kernels of hifi_gan/models.py:191-197, taken by the consumer)
  const float* x;
  const float* x2;
  const float* x3;
  float in_div;
  long long x_bs;
  int x_ld;
  // valid input length per batch row: in_len ? in_len[b] * in_mul : in_const
  const int* in_len;
  int in_mul;
  int in_const;
  // packed weights (see pack_conv_weights in weights_pack.h) and packed bias
  const float* w;
  const float* bias;
  int noct;  // octets per m-tile in the packed weights (ceil(Cin/8) rounded up to a multiple of 8)
  int Cin;
  int rows;  // number of valid virtual output rows
  int dil;
  int pad;
  float in_slope;  // leaky-relu slope applied to x on load (1.0 = identity)
  // output (rows < split) and second output (rows >= split, row index re-based)
  float* y;
  long long y_bs;
  int y_ld;
  const float* res;  // optional residual, same geometry as y
  float* y2;
  long long y2_bs;
  int y2_ld;
  int split;
  int accum;   // y  += instead of y  =
  int accum2;  // y2 += instead of y2 =
  float alpha;
  int out_act;
  // valid output length per batch row: out_len ? out_len[b] * out_mul : out_const
  const int* out_len;
  int out_mul;
  int out_const;
  // EPI_UPSAMPLE: stride and crop of the transposed conv; EPI_GATE/COUPLING: channels
  int up;
  int up_pad;
  int half;
  // EPI_COUPLING only, optional: InvConvNear (pre-inverted 4x4, n_split = 4) + ActNorm
  // reverse fused behind the coupling — z0 = first-half rows of the same tensor
  // EPI_GATE, multi-speaker voices: row b's speaker offsets of this layer (cond_layer(g) slice, layers.py:141-154), [2 half]
  // (tanh rows, then sigmoid rows) at cond + b * cond_bs; nullptr = none
  const float* cond;
  long long cond_bs;
  float* mix_x0;           // base of the flow tensor (first half), geometry of y
  const float* mix_w;      // [4][4] inverse weight
  const float* mix_bias;   // [2*half] ActNorm bias
  const float* mix_scale;  // [2*half] exp(-logs)
  // micro-benchmark ablations, honoured only by -DMI355TTS_ABLATION builds
  // (tools/conv_probe.py; results are WRONG when set): bit0 = no activation staging after
  // chunk 0, bit1 = no A-fragment loads after the prologue, bit2 = no per-chunk barrier
  int ablate;
  // workgroup -> tile order inside an XCD's run: 0 = row tile fastest (tiles of one time tile share their
  // input in one L2), 1 = time tile fastest (an XCD owns a range of row tiles = a slice of the weights)
  int rows_major;
  // split-bf16 mode (conv_bf16.h): the same weights as bf16 hi/lo fragments, and their slab count
  const void* w16;
  int nslab;
};

// XCD-aware tile order.  The dispatcher deals workgroup `lin` to XCD `lin % 8`, each XCD
// with its own L2.  Tiles that share input (the m-tiles of one time tile, and
// neighbouring time tiles through the halo) should meet in ONE L2, so the linear id is
// re-dealt: XCD x gets a contiguous run of tiles, m-tile fastest.  Bijective for any n
// (MI355X_MICROARCH.md, T1); a wrong placement guess costs speed, never correctness.
//
// `rows_major` != 0 flips the order inside the run: XCD x then owns a contiguous range of ROW tiles (with all
// their time tiles), i.e. 1/8 of the weights — for launches whose packed weights do not fit one 4 MB L2 while
// their input does (the stage-0 upsampler of HiFi-GAN 'high': 8.4 MB of weights, 1.3 MB of input; with time
// dealt across the XCDs every XCD streamed all 8.4 MB once per time tile: 86 MB fetched per launch).
__device__ __forceinline__ void xcd_tile_lin(int lin, int gx, int gy, int& tx, int& ty, int rows_major = 0) {
  const int n = gx * gy;
  const int xcd = lin & 7, slot = lin >> 3;
  const int q = n >> 3, r = n & 7;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  if (rows_major) {
    tx = id % gx;
    ty = id / gx;
  } else {
    ty = id % gy;
    tx = id / gy;
  }
}
__device__ __forceinline__ void xcd_tile(int gx, int gy, int& tx, int& ty, int rows_major = 0) {
  xcd_tile_lin(blockIdx.x + blockIdx.y * gx, gx, gy, tx, ty, rows_major);
}

// Ragged batches (gridDim.z > 1 rows of different lengths): the grid is sized for the longest row, and dealing
// contiguous runs of the GRID's tiles to the XCDs would hand a short row's few real tiles to XCD 0 (and 1) alone —
// over a batch of 8 rows with lengths 0.14 ... 1.0 of the longest, XCD 0 gets 8 shares of work and XCD 7 one
// (measured on BASELINE config 4: the 32-channel fused pair launches ran at 0.24 of peak against 0.59 at batch 1).
// So a row deals only ITS OWN tiles: the first gx_row * gy workgroups of the row's grid slice take them (spread
// evenly over the XCDs by the dispatcher's round-robin), the rest exit.
__device__ __forceinline__ int row_tiles(int n_len, int tile) { return (n_len + tile - 1) / tile; }

// extent of the implicit GEMM's N axis for batch row b (conv_tile derives the same value)
template <int K, int EPI>
__device__ __forceinline__ int conv_n_len(const ConvArgs& a, int b) {
  if constexpr (EPI == EPI_UPSAMPLE) {
    const int Lin = a.in_len ? a.in_len[b] * a.in_mul : a.in_const;
    return Lin > 0 ? Lin + (K - 1) : 0;
  } else {
    return a.out_len ? a.out_len[b] * a.out_mul : a.out_const;
  }
}

// LDS floats one workgroup of a tile shape needs (staging double buffer, reused by the k-group reduction)
template <int K, int CI_C, int MB, int NB, int WN, int KS, int HALO, int EPI, int WM = 1>
constexpr int conv_lds_floats() {
  constexpr bool SCATTER = (EPI == EPI_LINEAR || EPI == EPI_GATE);
  constexpr int NBR = (SCATTER && NB > 2) ? 1 : NB;
  constexpr int RED = (KS - 1) * WM * WN * NBR * 16 * 64;
  constexpr int XS = 2 * CI_C * (WN * NB * 32 + HALO);
  return XS > RED ? XS : RED;
}

// One workgroup's tile of the implicit GEMM: rows [32*MB*tile_y, +32*MB) x columns [T_T*tile_x, +T_T) of
// batch row b.  `xs` = conv_lds_floats<...>() floats of LDS.  Called by conv_mfma_kernel (one conv per
// launch) and by conv_group_kernel (the same-shaped convs of the three MRF chains in ONE launch).
template <int K, int CI_C, int MB, int NB, int WN, int KS, int HALO, int EPI, int WM = 1>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, const int tile_x, const int tile_y, const int b, float* __restrict__ xs) {
  // WM > 1 (EPI_LINEAR only): WM row groups of waves share ONE staged input tile — a workgroup then covers
  // WM*MB*32 output rows, so the input is staged once per WM m-tiles instead of once per m-tile.
  static_assert(WM == 1 || EPI == EPI_LINEAR || EPI == EPI_UPSAMPLE, "row groups of waves: linear and upsample epilogues");
  // Workgroup = WN x KS waves.  The WN waves of a k-group tile the time axis
  // (NB blocks of 32 columns each); the KS k-groups split the staged input
  // channels between them (octet o goes to group o % KS) and are summed through
  // LDS before the epilogue: two (or four) waves per SIMD from ONE staged tile,
  // which is what hides the LDS/L2 latencies at batch 1 where there are fewer
  // tiles than SIMDs.
  constexpr int NWAVES = WM * WN * KS;
  constexpr int WX = WM * WN;  // waves per k-group
  constexpr int NT = 64 * NWAVES;
  constexpr int T_T = WN * NB * 32;  // time columns per workgroup
  constexpr int XW = T_T + HALO;     // LDS row stride (floats)
  constexpr int OCTS = CI_C / 8;     // octets per staged chunk
  constexpr int NO = OCTS / KS;      // octets per chunk per k-group
  constexpr int S = NO * K;          // MFMA k-steps per chunk per k-group
  // the k-group reduction reuses the staging buffers as scratch; the reduce-scatter form
  // (LINEAR / GATE) may go NBR column blocks per round to bound it
  constexpr bool SCATTER = (EPI == EPI_LINEAR || EPI == EPI_GATE);
  constexpr int NBR = (SCATTER && NB > 2) ? 1 : NB;
  constexpr int RED = (KS - 1) * WX * NBR * 16 * 64;
  constexpr int LDSF = conv_lds_floats<K, CI_C, MB, NB, WN, KS, HALO, EPI, WM>();
  static_assert(CI_C % 8 == 0 && CI_C % NWAVES == 0 && OCTS % KS == 0, "bad tile parameters");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = (wave / WM) % WN;
  const int kg = wave / WX;
  const int wx = wm * WN + wn;  // this wave's slot inside its k-group
  const int t0 = tile_x * T_T;
  const int mt0 = (tile_y * WM + wm) * MB;

  const int Lin = a.in_len ? a.in_len[b] * a.in_mul : a.in_const;
  const int Lout = a.out_len ? a.out_len[b] * a.out_mul : a.out_const;
  // extent of the GEMM's N axis for this batch row
  const int n_len = (EPI == EPI_UPSAMPLE) ? (Lin > 0 ? Lin + (K - 1) : 0) : Lout;
  if (t0 >= n_len) return;  // uniform per workgroup

  const int roww = T_T + (K - 1) * a.dil;  // staged columns actually used
  const float slope = a.in_slope;
  const float* xb = a.x + (long long)b * a.x_bs;
  const float* xb2 = a.x2 ? a.x2 + (long long)b * a.x_bs : nullptr;
  const float* xb3 = a.x3 ? a.x3 + (long long)b * a.x_bs : nullptr;
  // whole chunks covering the real channels; the packed weights are zero-padded past Cin
  const int nchunks = ((a.Cin + 7) / 8 + OCTS - 1) / OCTS;

  // ---- staging of the activation tile (global -> VGPR -> LDS), 16 bytes per lane:
  // the tile starts at the 4-aligned column t0 - PA (PA = pad rounded up to 4), every
  // lane loads whole float4s from clamped in-range addresses and zeroes the
  // out-of-range elements by select (no branches, loads issue back to back), and
  // lands them with ds_write_b128.  Row strides (x_ld) are multiples of 4 floats.
  // Two register sets: the tile for chunk c+2 is requested while chunk c computes,
  // so a staged chunk has two MFMA phases (not one) to cover the L2/MALL latency.
  constexpr int XW4 = XW / 4;
  constexpr int NF4 = CI_C * XW4;             // float4s per staged chunk
  constexpr int NE = (NF4 + NT - 1) / NT;     // per thread, flattened over (row, float4) so no lane idles
  static_assert(XW % 4 == 0, "LDS row stride must be a multiple of 4 floats");
  // Prefetch distance: two chunks (two register sets) by default; the 64-row one-time-wave
  // tiles with wide halos (k >= 7: 4 float4 per thread per chunk) prefetch one chunk ahead
  // with a single set, which keeps them under 128 VGPRs (two workgroups per CU) unspilled.
  constexpr bool PD1 = (WN == 1 && KS == 8 && MB == 2 && K >= 7);
  float4 preA[NE], preB[PD1 ? 1 : NE];
  const int PA = (a.pad + 3) & ~3;
  const int used4 = (PA - a.pad + roww + 3) >> 2;  // float4s per row actually needed
  const int cin_last = a.Cin - 1;
  const int ld_last4 = a.x_ld - 4;
  // gload only ISSUES the loads of a chunk (raw values stay in flight in `pre`); the
  // masking and the input activation are applied in lstore, one chunk of MFMA work later,
  // when the data has long arrived.  (Doing them in gload made every wave wait for its
  // own prefetch at the top of each chunk — a full memory round trip per chunk.)
  auto gload = [&](int chunk, float4 (&pre)[NE]) {
    int off[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / XW4, f = e - row * XW4;
      const int ci = chunk * CI_C + (row < CI_C ? row : CI_C - 1);
      const int c0 = t0 - PA + 4 * f;
      off[i] = (ci < a.Cin ? ci : cin_last) * a.x_ld + (c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0));
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) pre[i] = *reinterpret_cast<const float4*>(xb + off[i]);
#ifndef MI355TTS_PROBE_NO_X2  // (probe builds only: the staging path without the MRF-average inputs)
    if (xb2) {  // wave-uniform: MRF average of the previous stage's chains, batched (this path waits)
      float4 t2[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) t2[i] = *reinterpret_cast<const float4*>(xb2 + off[i]);
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        pre[i].x += t2[i].x;
        pre[i].y += t2[i].y;
        pre[i].z += t2[i].z;
        pre[i].w += t2[i].w;
      }
      if (xb3) {
#pragma unroll
        for (int i = 0; i < NE; ++i) t2[i] = *reinterpret_cast<const float4*>(xb3 + off[i]);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
          pre[i].x += t2[i].x;
          pre[i].y += t2[i].y;
          pre[i].z += t2[i].z;
          pre[i].w += t2[i].w;
        }
      }
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        pre[i].x = pre[i].x / a.in_div;
        pre[i].y = pre[i].y / a.in_div;
        pre[i].z = pre[i].z / a.in_div;
        pre[i].w = pre[i].w / a.in_div;
      }
    }
#endif
  };
  auto lstore = [&](int buf, int chunk, const float4 (&pre)[NE]) {
    float4* dst = reinterpret_cast<float4*>(xs + buf * (CI_C * XW));
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / XW4, f = e - row * XW4;
      const int c0 = t0 - PA + 4 * f;
      const bool fok = chunk * CI_C + row < a.Cin && f < used4;
      float4 v = pre[i];
      v.x = (fok && c0 >= 0 && c0 < Lin) ? v.x : 0.f;
      v.y = (fok && c0 + 1 >= 0 && c0 + 1 < Lin) ? v.y : 0.f;
      v.z = (fok && c0 + 2 >= 0 && c0 + 2 < Lin) ? v.z : 0.f;
      v.w = (fok && c0 + 3 >= 0 && c0 + 3 < Lin) ? v.w : 0.f;
      v.x = v.x > 0.f ? v.x : v.x * slope;
      v.y = v.y > 0.f ? v.y : v.y * slope;
      v.z = v.z > 0.f ? v.z : v.z * slope;
      v.w = v.w > 0.f ? v.w : v.w * slope;
      if (e < NF4) dst[e] = v;  // rows are XW = 4*XW4 floats: the flat float4 index IS the LDS index
    }
  };

  floatx16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  // A-fragment stream of m-tile mt: float4 index ((mt*noct + oct)*K + k)*64 + lane.
  // This k-group walks steps q = 0..S-1 per chunk: octet kg + (q/K)*KS, tap q%K.
  const float4* wq[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
    wq[mb] = reinterpret_cast<const float4*>(a.w) + (long long)(mt0 + mb) * a.noct * K * 64 + lane;
  const int last_chunk = nchunks - 1;
  const int last_step = nchunks * S - 1;
  auto a_index = [&](int chunk, int q) -> long long {
    // q may run past this chunk (prefetch): roll into the following ones, clamp at the end
    int g = chunk * S + q;
    g = g < last_step ? g : last_step;
    const int ch = g / S;
    q = g - ch * S;
    const int oi = q / K;
    const int k = q - oi * K;
    return (long long)(((ch * OCTS + kg + oi * KS) * K + k)) * 64;
  };

  // RD-deep register ring of A fragments: step q uses ar[q % RD] while the loads
  // for steps q+1 .. q+RD-1 are in flight (L2 latency ~ one MFMA step)
  constexpr int RD = MI355TTS_ARING;
  float4 ar[RD][MB];

  // prologue: everything the first MFMA needs goes out in ONE batch of loads — the
  // first activation chunk, the first weight fragments (cold in L2: a layer's weights are
  // read for the first time here), then the second chunk — before anything is waited for
  CONV_STAMP(0);
  gload(0, preA);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < RD - 1; ++i) ar[i][mb] = wq[mb][a_index(0, i)];
  if constexpr (!PD1) {
    if (nchunks > 1) gload(1, preB);
  }
  CONV_STAMP(1);
  lstore(0, 0, preA);
  __syncthreads();
  CONV_STAMP(2);

  const int b_off = (lane >> 5) * XW + wn * (NB * 32) + (lane & 31) + (PA - a.pad);

  auto do_chunk = [&](int chunk, float4 (&pre_load)[NE], const float4 (&pre_store)[NE]) {
    const int buf = chunk & 1;
    const bool more = chunk < last_chunk;
    CONV_CHUNK_STAMP(chunk, 0);
    if constexpr (PD1) {
      if (more) gload(chunk + 1, pre_load);
    } else {
      if (chunk + 2 < nchunks && !MI355TTS_ABLATE(a, 1)) gload(chunk + 2, pre_load);
    }
    const float* xt = xs + buf * (CI_C * XW) + b_off + kg * 8 * XW;
    float bcur[4][NB], bnxt[4][NB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bcur[j][nb] = xt[(2 * j) * XW + nb * 32];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      // the loads for later steps and this step's MFMAs (interleaved by the hints below)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        if (!MI355TTS_ABLATE(a, 2)) ar[(s + RD - 1) % RD][mb] = wq[mb][a_index(chunk, s + RD - 1)];
      if (s + 1 < S) {
        const int oi = (s + 1) / K;
        const int k = (s + 1) - oi * K;
        const float* bp = xt + (oi * KS * 8) * XW + k * a.dil;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bnxt[j][nb] = bp[(2 * j) * XW + nb * 32];
      }
      // With two or more accumulators per wave, consecutive MFMAs are independent and the
      // fillers are interleaved with them (below).  With ONE accumulator every MFMA depends
      // on the previous one, and a filler between two dependent MFMAs costs ~40 cycles
      // (MI355X_MICROARCH.md): there the fillers go first and the MFMAs stay back to back.
      constexpr bool INTERLEAVE = MB * NB >= 2;
      if constexpr (!INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float4 af = ar[s % RD][mb];
          const float av = (j == 0) ? af.x : (j == 1) ? af.y : (j == 2) ? af.z : af.w;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bcur[j][nb], acc[mb][nb], 0, 0, 0);
        }
      }
      // Issue order inside the step (sched_group_barrier: 0x008 = MFMA, 0x020 = VMEM read,
      // 0x100 = LDS read): each filler — the fragment loads for step s+RD-1, the LDS reads
      // for step s+1 — goes right behind an MFMA, so it issues in the shadow of that MFMA's
      // 64 cycles instead of in a gap before the burst (−1…−3 % on the ResBlock convs in the
      // pipeline trace, up to −9 % in the per-layer sweep).
      if constexpr (INTERLEAVE) {
#pragma unroll
        for (int i = 0; i < 4 * MB * NB; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < MB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          else if (i < MB + 2 * NB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (s + 1 < S) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bcur[j][nb] = bnxt[j][nb];
      }
    }
    // re-base the ring for the next chunk: its steps 0..RD-2 sit in slots (S+i) % RD
    if constexpr (S % RD != 0) {
      float4 rr[RD - 1][MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < RD - 1; ++i) rr[i][mb] = ar[(S + i) % RD][mb];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int i = 0; i < RD - 1; ++i) ar[i][mb] = rr[i][mb];
    }
    CONV_CHUNK_STAMP(chunk, 1);
    if (more && !MI355TTS_ABLATE(a, 1)) lstore(buf ^ 1, chunk + 1, pre_store);
    CONV_CHUNK_STAMP(chunk, 2);
    if (!MI355TTS_ABLATE(a, 4)) __syncthreads();
    CONV_CHUNK_STAMP(chunk, 3);
  };
  if constexpr (PD1) {
    for (int chunk = 0; chunk < nchunks; ++chunk) do_chunk(chunk, preA, preA);
  } else {
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      do_chunk(chunk, preA, preB);
      if (chunk + 1 < nchunks) do_chunk(chunk + 1, preB, preA);
    }
  }

  CONV_STAMP(3);
  // ---------------------------------------------------------------- k-group reduction
  // C/D map of v_mfma_f32_32x32x2_f32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  constexpr int R = 16 / KS;  // accumulator registers per 32x32 block that one k-group ends up owning
  // Register g-th group owns (rr = 0..R-1).  LINEAR: R consecutive registers.  GATE pairs
  // block rows i and i+16 = registers r and r+8 of one lane, so a group owns R/2 low
  // registers and their partners.
  auto reg_of = [](int g, int rr) constexpr -> int {
    if constexpr (EPI == EPI_GATE) return rr < R / 2 ? g * (R / 2) + rr : 8 + g * (R / 2) + (rr - R / 2);
    else return g * R + rr;
  };
  float own[MB][NB][R];
  if constexpr (SCATTER) {
    // Reduce-scatter: the KS partial tiles are summed through LDS so that EVERY k-group
    // ends up with the finished values of 1/KS of the registers, and all WN x KS waves
    // share the epilogue's loads and stores (a reduce-to-group-0 leaves KS-1 of every KS
    // waves idle through the longest memory round trips of the kernel).  The staging
    // buffers are free: the loop ended on a barrier.
#pragma unroll
    for (int g = 0; g < KS; ++g)
      if (kg == g) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int rr = 0; rr < R; ++rr) own[mb][nb][rr] = acc[mb][nb][reg_of(g, rr)];
      }
    if constexpr (KS > 1) {
      // scratch per round: owners x sources x WN x NBR blocks x R registers x 64 lanes
      static_assert(NB % NBR == 0 && RED <= LDSF, "reduction scratch must fit in the staging buffers");
      float* red = xs;
      bool first_round = true;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb0 = 0; nb0 < NB; nb0 += NBR) {
          if (!first_round) __syncthreads();
          first_round = false;
#pragma unroll
          for (int g = 0; g < KS; ++g)
            if (kg != g) {
              const int si = kg < g ? kg : kg - 1;
#pragma unroll
              for (int nbi = 0; nbi < NBR; ++nbi)
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                  red[((((g * (KS - 1) + si) * WX + wx) * NBR + nbi) * R + rr) * 64 + lane] = acc[mb][nb0 + nbi][reg_of(g, rr)];
            }
          __syncthreads();
#pragma unroll
          for (int si = 0; si < KS - 1; ++si)
#pragma unroll
            for (int nbi = 0; nbi < NBR; ++nbi)
#pragma unroll
              for (int rr = 0; rr < R; ++rr)
                own[mb][nb0 + nbi][rr] += red[((((kg * (KS - 1) + si) * WX + wx) * NBR + nbi) * R + rr) * 64 + lane];
        }
    }
  } else if constexpr (KS > 1) {
    // sum the k-groups' partial tiles through LDS, one m-block per round; group 0 then
    // owns the epilogue (COUPLING / UPSAMPLE: their epilogues need whole register sets)
    float* red = xs;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if (mb > 0) __syncthreads();
      if (kg > 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[((((kg - 1) * WX + wx) * NB + nb) * 16 + r) * 64 + lane] = acc[mb][nb][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int g = 1; g < KS; ++g)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[mb][nb][r] += red[((((g - 1) * WX + wx) * NB + nb) * 16 + r) * 64 + lane];
      }
    }
    if (kg > 0) return;
  }

  CONV_STAMP(4);
  // ---------------------------------------------------------------- epilogue
  // Loads (bias / residual / accumulate) go out in batches from clamped,
  // always-valid addresses under wave-uniform conditions only; lanes outside the
  // tensor just skip the store.  (One exec-masked branch per element would cost a
  // full memory round trip per element.)
  const int col = lane & 31;
  const int rbase = 4 * (lane >> 5);

  if constexpr (EPI == EPI_LINEAR) {
    // this wave finishes registers reg_of(kg, 0..R-1) of each of its blocks
    int rowin[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      const int r = kg * R + rr;
      rowin[rr] = (r & 3) + 8 * (r >> 2) + rbase;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int row0 = (mt0 + mb) * 32;
      const bool first = row0 < a.split;  // split is a multiple of 32 (or 0 / >= rows)
      float* yp = first ? a.y : a.y2;
      const long long ybs = first ? a.y_bs : a.y2_bs;
      const int yld = first ? a.y_ld : a.y2_ld;
      const int rowoff = first ? 0 : a.split;
      const float* rp = first ? a.res : nullptr;
      const bool acc_on = first ? (a.accum != 0) : (a.accum2 != 0);
      const float alpha = first ? a.alpha : 1.0f;
      const int act = first ? a.out_act : (int)ACT_NONE;
      float bb[R];
      int roff[R];
      bool rok[R];
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        const int row = row0 + rowin[rr];
        bb[rr] = a.bias ? a.bias[row] : 0.f;  // packed bias is padded to whole m-tiles
        rok[rr] = row < a.rows;
        roff[rr] = ((rok[rr] ? row : a.rows - 1) - rowoff) * yld;
      }
      float v[NB][R];
      int tcol[NB];
      bool tok[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int t = t0 + (wn * NB + nb) * 32 + col;
        tok[nb] = t < Lout;
        tcol[nb] = tok[nb] ? t : Lout - 1;
#pragma unroll
        for (int rr = 0; rr < R; ++rr) v[nb][rr] = own[mb][nb][rr] + bb[rr];
      }
      float* yb = yp + (long long)b * ybs;
      if (rp) {
        const float* rb = rp + (long long)b * ybs;
        float rv[NB][R];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) rv[nb][rr] = rb[roff[rr] + tcol[nb]];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) v[nb][rr] += rv[nb][rr];
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) v[nb][rr] *= alpha;
      if (acc_on) {
        float ov[NB][R];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) ov[nb][rr] = yb[roff[rr] + tcol[nb]];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) v[nb][rr] += ov[nb][rr];
      }
      if (act == ACT_RELU) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) v[nb][rr] = v[nb][rr] > 0.f ? v[nb][rr] : 0.f;
      } else if (act == ACT_TANH) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int rr = 0; rr < R; ++rr) v[nb][rr] = tanhf(v[nb][rr]);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
          if (rok[rr] && tok[nb]) yb[roff[rr] + tcol[nb]] = v[nb][rr];
    }
  } else if constexpr (EPI == EPI_GATE) {
    static_assert(MB == 1, "paired epilogues use one 32-row tile: 16 rows of each half");
    // virtual tile p = blockIdx.y holds rows c = 16p + i (i < 16) of the first half
    // (tanh) in block rows 0..15 and the matching rows of the second half (sigmoid) in
    // block rows 16..31.  In the C/D map, block row i and row i + 16 sit in the SAME
    // lane, registers r and r + 8 — the pair meets in registers; this wave owns the low
    // registers kg*H .. kg*H+H-1 and their partners (own[..][H + rr]).
    constexpr int H = R / 2;
    static_assert(H >= 1, "GATE needs at most 8 k-groups");
    float b0[H], b1[H];
    int off[H];
    bool cok[H];
#pragma unroll
    for (int rr = 0; rr < H; ++rr) {
      const int r = kg * H + rr;
      const int i = (r & 3) + 8 * (r >> 2) + rbase;  // 0..15
      b0[rr] = a.bias ? a.bias[mt0 * 32 + i] : 0.f;
      b1[rr] = a.bias ? a.bias[mt0 * 32 + 16 + i] : 0.f;
      const int c = tile_y * 16 + i;
      cok[rr] = c < a.half;
      if (a.cond) {  // x_in + g_l (layers.py:154): the speaker's offsets of this layer
        const float* cd = a.cond + (long long)b * a.cond_bs + (cok[rr] ? c : a.half - 1);
        b0[rr] += cd[0];
        b1[rr] += cd[a.half];
      }
      off[rr] = (cok[rr] ? c : a.half - 1) * a.y_ld;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int t = t0 + (wn * NB + nb) * 32 + col;
      const bool tok = t < Lout;
      float* yb = a.y + (long long)b * a.y_bs + (tok ? t : Lout - 1);
#pragma unroll
      for (int rr = 0; rr < H; ++rr) {
        const float v0 = own[0][nb][rr] + b0[rr];
        const float v1 = own[0][nb][H + rr] + b1[rr];
        const float out = tanhf(v0) * (1.0f / (1.0f + expf(-v1)));
        if (cok[rr] && tok) yb[off[rr]] = out;
      }
    }
  } else if constexpr (EPI == EPI_COUPLING) {
    static_assert(MB == 1, "paired epilogues use one 32-row tile: 16 rows of each half");
    // virtual tile p = blockIdx.y holds rows c = 16p + i (i < 16) of the first half
    // (tanh / m) in block rows 0..15 and the matching rows of the second half
    // (sigmoid / logs) in block rows 16..31.  In the C/D map, block row i and row
    // i + 16 sit in the SAME lane, registers r and r + 8 — the pair meets in registers.
    float b0[8], b1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + rbase;  // 0..15
      b0[r] = a.bias ? a.bias[mt0 * 32 + i] : 0.f;
      b1[r] = a.bias ? a.bias[mt0 * 32 + 16 + i] : 0.f;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int t = t0 + (wn * NB + nb) * 32 + col;
      const bool tok = t < Lout;
      const int tc = tok ? t : Lout - 1;
      float* yb = a.y + (long long)b * a.y_bs + tc;
      int off[8];
      bool ok[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = tile_y * 16 + (r & 3) + 8 * (r >> 2) + rbase;
        const bool cok = c < a.half;
        ok[r] = cok && tok;
        off[r] = (cok ? c : a.half - 1) * a.y_ld;
      }
      float out[8];
      {
        const float* rb = a.res + (long long)b * a.y_bs + tc;
        float rv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) rv[r] = rb[off[r]];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float v0 = acc[0][nb][r] + b0[r];
          const float v1 = acc[0][nb][r + 8] + b1[r];
          out[r] = (rv[r] - v0) * expf(-v1);
        }
        if (a.mix_w) {
          // InvConvNear reverse + ActNorm reverse (glow_tts/layers.py:238-272, 192-194) on the
          // channel groups {2k, 2k+1, half+2k, half+2k+1}: registers (r, r+1), r even, hold the
          // freshly coupled second-half pair; the first-half pair is fetched, all four are
          // mixed with the pre-inverted 4x4 and written back normalised.
          float* x0b = a.mix_x0 + (long long)b * a.y_bs + tc;
          float w[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) w[i] = a.mix_w[i];
          float xa[4], xb2[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            xa[q] = x0b[off[2 * q]];
            xb2[q] = x0b[off[2 * q + 1]];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 2 * q;
            const int c0 = tile_y * 16 + (r & 3) + 8 * (r >> 2) + rbase;  // even channel 2k
            const int cc = c0 < a.half ? c0 : a.half - 2;
            const float in0 = xa[q], in1 = xb2[q], in2 = out[r], in3 = out[r + 1];
            float o[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) o[m] = w[m * 4 + 0] * in0 + w[m * 4 + 1] * in1 + w[m * 4 + 2] * in2 + w[m * 4 + 3] * in3;
            const float y0 = (o[0] - a.mix_bias[cc]) * a.mix_scale[cc];
            const float y1 = (o[1] - a.mix_bias[cc + 1]) * a.mix_scale[cc + 1];
            out[r] = (o[2] - a.mix_bias[a.half + cc]) * a.mix_scale[a.half + cc];
            out[r + 1] = (o[3] - a.mix_bias[a.half + cc + 1]) * a.mix_scale[a.half + cc + 1];
            if (ok[r]) {
              x0b[off[r]] = y0;
              x0b[off[r + 1]] = y1;
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (ok[r]) yb[off[r]] = out[r];
    }
  } else {  // EPI_UPSAMPLE
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float bb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bb[r] = a.bias ? a.bias[(mt0 + mb) * 32 + (r & 3) + 8 * (r >> 2) + rbase] : 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int q = t0 + (wn * NB + nb) * 32 + col;
        if (q >= n_len) continue;
        if ((a.up & 3) == 0 && (a.up_pad & 3) == 0 && (a.y_ld & 3) == 0) {
          // registers 4g..4g+3 of a lane are 4 consecutive virtual rows = 4 consecutive phases of ONE output
          // channel = 4 consecutive, 16-byte aligned output samples: one float4 store instead of four scalar
          // stores 4*up bytes apart (the two half-waves fill the other phases of the same 32-byte runs)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int row0 = (mt0 + mb) * 32 + 8 * g4 + rbase;
            if (row0 >= a.rows) continue;
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
          }
          continue;
        }
#ifndef MI355TTS_NO_UP2_PAIR  // (A/B builds only)
        if (a.up == 2) {
          // stride 2: registers (r, r + 1), r even, of a lane are the two phases of ONE output channel = two consecutive
          // output samples; the lanes of a half-wave are consecutive q, so one 8-byte store per pair writes 256
          // contiguous bytes per half-wave instead of two interleaved 4-byte stores at stride 8.  (4-byte aligned only:
          // up_pad is odd for the k = 4 upsamplers; the hardware takes dword-aligned multi-dword stores.)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int row = (mt0 + mb) * 32 + (r & 3) + 8 * (r >> 2) + rbase;  // even
            if (row >= a.rows) continue;
            const int co = row >> 1;
            const int n0 = q * 2 - a.up_pad;
            float* dst = a.y + (long long)b * a.y_bs + (long long)co * a.y_ld + n0;
            const float v0 = acc[mb][nb][r] + bb[r], v1 = acc[mb][nb][r + 1] + bb[r + 1];
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
#endif
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (mt0 + mb) * 32 + (r & 3) + 8 * (r >> 2) + rbase;
          if (row >= a.rows) continue;
          const int co = row / a.up;
          const int ph = row - co * a.up;
          const int n = q * a.up + ph - a.up_pad;
          if (n < 0 || n >= Lout) continue;
          a.y[(long long)b * a.y_bs + (long long)co * a.y_ld + n] = acc[mb][nb][r] + bb[r];
        }
      }
    }
  }
}


template <int K, int CI_C, int MB, int NB, int WN, int KS, int HALO, int EPI, int WM = 1>
// (one-column-block variants sit at 110-135 VGPRs: ask for <= 128 so two workgroups share a CU)
__global__ __launch_bounds__(64 * WM * WN * KS, ((EPI == EPI_LINEAR && ((NB == 1 && MB == 2) || (MB == 1 && NB == 2 && KS == 4))) || WM == 4) ? 4 : 1) void conv_mfma_kernel(const ConvArgs a) {
  __shared__ float xs[conv_lds_floats<K, CI_C, MB, NB, WN, KS, HALO, EPI, WM>()];
  int tile_x, tile_y;
  int gx = gridDim.x;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  if (gridDim.z > 1) {  // ragged batch: this row's own tiles only (see row_tiles)
    gx = row_tiles(conv_n_len<K, EPI>(a, blockIdx.z), WN * NB * 32);
    if (lin >= gx * (int)gridDim.y) return;
  }
  xcd_tile_lin(lin, gx, gridDim.y, tile_x, tile_y, a.rows_major);
  conv_tile<K, CI_C, MB, NB, WN, KS, HALO, EPI, WM>(a, tile_x, tile_y, blockIdx.z, xs);
  CONV_STAMP(5);
}

// The MRF chains of a HiFi-GAN stage (hifi_gan/models.py:191-197) run convs of the SAME geometry
// (channels, length, tile shape) that differ only in tap count — K0 >= K1 >= K2 — dilation and
// weights.  At batch 1 one such conv cannot fill the chip (a few hundred to ~1200 workgroups of
// ~10 us: launch ramp, tail quantisation and every workgroup's prologue/epilogue are exposed), so
// the three are issued as ONE launch: a 1-D grid whose first n0 workgroups are conv 0's tiles
// (the longest-running ones first), the next n1 conv 1's, the rest conv 2's.  Each tile runs
// exactly the code of conv_mfma_kernel, so results are bit-identical to three launches.
// Group sizes are padded to multiples of 8 so a tile's XCD (workgroup id % 8) is the one
// xcd_tile_lin assumes.
constexpr int GROUP_MAX_SEG = 8;
struct ConvGroupArgs {
  ConvArgs c[3];
  int gx[3], gy[3];  // tile grid of each member (x = time tiles, y = row tiles)
  int off[4];        // first workgroup of each member (multiples of 8), off[3] = grid size
  // Optional dispatch order (rb_group_kernel; group_snake_order below): workgroups [seg_off[s], seg_off[s + 1]) run tiles
  // seg_first[s] ... of member seg_m[s]; every bound a multiple of 8, so a tile's XCD stays workgroup id % 8.  nseg = 0: the
  // plain longest-first order of `off`.
  int nseg = 0;
  int seg_off[GROUP_MAX_SEG + 1];
  int seg_first[GROUP_MAX_SEG];
  int seg_m[GROUP_MAX_SEG];
};

// A grouped launch whose workgroups are ALL resident at once (<= `slots` = CUs x workgroups per CU) is not balanced by the
// dispatcher: it deals workgroup i to CU i mod `ncu` (XCD i mod 8, then round-robin over the XCD's CUs — measured, every one of
// 224 second-round workgroups of a 480-workgroup launch sat on the CU of workgroup i - 256: profiles/NOTES.md), so with the
// longest-first order the CUs that got an 11-tap tile in the first round get the 7-tap tiles of the second, and the launch
// lasts 18 tap-units where the mean CU has 12.8.  This lays the rounds out as a snake — round 0 longest first, round 1
// SHORTEST first, ... — so the CU with the longest tile of one round gets the shortest of the next (14 tap-units for that
// launch).  Launches with more workgroups than slots keep the plain order (the dispatcher then balances them as slots free).
inline void group_snake_order(ConvGroupArgs& g, int ncu, int slots) {
  g.nseg = 0;
  const int total = g.off[3];
  if (ncu <= 0 || (ncu & 7) || total <= ncu || total > slots) return;
  struct Piece { int m, first, n; };
  Piece out[GROUP_MAX_SEG];
  int nout = 0;
  for (int r0 = 0; r0 < total; r0 += ncu) {
    const int r1 = r0 + ncu < total ? r0 + ncu : total;
    Piece round[3];
    int nr = 0;
    for (int m = 0; m < 3; ++m) {  // the part of member m (workgroups off[m] .. off[m + 1] of the plain order) inside this round
      const int a = g.off[m] > r0 ? g.off[m] : r0, b = g.off[m + 1] < r1 ? g.off[m + 1] : r1;
      if (b > a) round[nr++] = Piece{m, a - g.off[m], b - a};
    }
    const bool rev = ((r0 / ncu) & 1) != 0;
    for (int i = 0; i < nr; ++i) {
      const Piece& pc = round[rev ? nr - 1 - i : i];
      if (nout && out[nout - 1].m == pc.m && out[nout - 1].first + out[nout - 1].n == pc.first) {
        out[nout - 1].n += pc.n;
        continue;
      }
      if (nout == GROUP_MAX_SEG) return;  // (cannot happen with three members and <= 4 rounds; keep the plain order)
      out[nout++] = pc;
    }
  }
  int o = 0;
  for (int i = 0; i < nout; ++i) {
    g.seg_off[i] = o;
    g.seg_first[i] = out[i].first;
    g.seg_m[i] = out[i].m;
    o += out[i].n;
  }
  g.seg_off[nout] = o;
  g.nseg = nout;
}

// What that dispatch order does to the busiest CU: (largest per-CU sum of tile costs) / (mean per-CU sum), for `real[m]` real
// tiles per member (the rest of a member's range is padding that exits at once) of cost `cost[m]` each, workgroup i on CU
// i mod ncu.  1.0 = perfectly even.  promote_group_plans (host_launch.h) moves a step to the big tile only when this is small:
// just above one workgroup per CU the big tile's few extra workgroups double the busiest CUs' work.
inline double group_order_imbalance(const ConvGroupArgs& g, int ncu, const int real[3], const double cost[3]) {
  if (ncu <= 0 || ncu > 4096) return 1e9;
  double load[4096];
  for (int c = 0; c < ncu; ++c) load[c] = 0.0;
  double total = 0.0;
  auto place = [&](int lin, int m, int l) {
    if (l >= real[m]) return;
    load[lin % ncu] += cost[m];
    total += cost[m];
  };
  if (g.nseg) {
    for (int sg = 0; sg < g.nseg; ++sg)
      for (int lin = g.seg_off[sg]; lin < g.seg_off[sg + 1]; ++lin) place(lin, g.seg_m[sg], g.seg_first[sg] + (lin - g.seg_off[sg]));
  } else {
    for (int m = 0; m < 3; ++m)
      for (int lin = g.off[m]; lin < g.off[m + 1]; ++lin) place(lin, m, lin - g.off[m]);
  }
  double mx = 0.0;
  for (int c = 0; c < ncu; ++c) mx = load[c] > mx ? load[c] : mx;
  return total > 0.0 ? mx / (total / ncu) : 1e9;
}
template <int K0, int K1, int K2, int CI_C, int MB, int NB, int WN, int KS, int H0, int H1, int H2, int WM = 1>
// (second launch bound = waves per SIMD: 4 keeps every member under 128 VGPRs, i.e. two 8-wave workgroups per CU)
__global__ __launch_bounds__(64 * WM * WN * KS, ((NB == 1 && MB == 2) || WM == 4 || (MB == 1 && NB == 2 && KS == 4)) ? 4 : 1) void conv_group_kernel(const ConvGroupArgs g) {
  constexpr int L0 = conv_lds_floats<K0, CI_C, MB, NB, WN, KS, H0, EPI_LINEAR, WM>();
  constexpr int L1 = conv_lds_floats<K1, CI_C, MB, NB, WN, KS, H1, EPI_LINEAR, WM>();
  constexpr int L2 = conv_lds_floats<K2, CI_C, MB, NB, WN, KS, H2, EPI_LINEAR, WM>();
  __shared__ float xs[L0 > L1 ? (L0 > L2 ? L0 : L2) : (L1 > L2 ? L1 : L2)];
  const int lin = blockIdx.x;
  const int b = blockIdx.z;
  const bool ragged = gridDim.z > 1;  // rows of different lengths: a row deals only its own tiles (see row_tiles)
  constexpr int T_T = WN * NB * 32;
  int tx, ty;
  CONV_WG_STAMP(lin, 0);
  if (lin < g.off[1]) {
    const int gx = ragged ? row_tiles(conv_n_len<K0, EPI_LINEAR>(g.c[0], b), T_T) : g.gx[0];
    if (lin >= gx * g.gy[0]) return;
    xcd_tile_lin(lin, gx, g.gy[0], tx, ty);
    conv_tile<K0, CI_C, MB, NB, WN, KS, H0, EPI_LINEAR, WM>(g.c[0], tx, ty, b, xs);
  } else if (lin < g.off[2]) {
    const int l = lin - g.off[1];
    const int gx = ragged ? row_tiles(conv_n_len<K1, EPI_LINEAR>(g.c[1], b), T_T) : g.gx[1];
    if (l >= gx * g.gy[1]) return;
    xcd_tile_lin(l, gx, g.gy[1], tx, ty);
    conv_tile<K1, CI_C, MB, NB, WN, KS, H1, EPI_LINEAR, WM>(g.c[1], tx, ty, b, xs);
  } else {
    const int l = lin - g.off[2];
    const int gx = ragged ? row_tiles(conv_n_len<K2, EPI_LINEAR>(g.c[2], b), T_T) : g.gx[2];
    if (l >= gx * g.gy[2]) return;
    xcd_tile_lin(l, gx, g.gy[2], tx, ty);
    conv_tile<K2, CI_C, MB, NB, WN, KS, H2, EPI_LINEAR, WM>(g.c[2], tx, ty, b, xs);
  }
  CONV_WG_STAMP(lin, 1);
}

}  // namespace mi355tts
