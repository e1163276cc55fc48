// The WaveNet gate conv of the GlowTTS decoder (glow_tts/layers.py:138-162: in_layers[i], k taps, H -> 2H rows, then
// tanh(first half) * sigmoid(second half), glow_tts/utils.py:31-38) on 16-row x 32-column tiles.
//
// Why a second tile for one layer type: at batch 1 the decoder's time axis is ~312 columns.  conv_mfma.h's smallest
// tile (32 virtual rows x 32 columns, 8 k-groups) yields 12 x 10 = 120 workgroups for H = 192 — under half of the 256
// CUs — and each of them holds 480 v_mfma_f32_32x32x2 on ONE CU's four matrix pipes: 7680 cycles per SIMD, 3.2-3.8 us,
// which is what its main loop measures (4.9 of the launch's 9.1 us; tools/probe/glow_conv_bench.hip).  More k-groups
// do not help (the pipes are the CU's), smaller tiles do: v_mfma_f32_16x16x4_f32 has the same FLOP rate per pipe, so a
// 16-row tile = 8 gate channels (8 tanh rows + their 8 sigmoid rows) x 32 columns is 240 workgroups of half the work.
//
// One workgroup = 8 waves = 8 k-groups.  All Cin input channels of the tile's 32 (+ halo) columns are staged once
// (f32, [channel][48 columns]: the stride puts the four K-lanes of a B fragment in four disjoint bank groups);
// k-group g takes the 4-channel groups g, g + 8, ... with all their taps, its A fragments (pre-packed per lane, one
// dword per MFMA) all requested at entry; two accumulators (column blocks) per wave make consecutive MFMAs
// independent.  The 8 partial tiles meet in LDS, where thread (channel i, column n) finds its tanh and sigmoid rows.
//
// lin16_kernel (second half of this file) is the same tile with a plain epilogue for the other small-launch convs of
// GlowTTS (FFN, duration predictor, prenet, 1 x 1 convs), optionally with the producer's LayerNorm as a prologue.
#pragma once
#include <hip/hip_runtime.h>
#include "prio.h"

namespace mi355tts {

#ifndef GATE_STAMP
#define GATE_STAMP(n)
#endif

typedef float gate_floatx4 __attribute__((ext_vector_type(4)));

// Staged halo + alignment columns per channel row.  16 made the row stride 48 = 16 mod 32 (conflict-free fragment reads) but the
// tile of a 192-channel conv 36 KB — one LDS page MORE than the 32 KB hole a finishing ResBlock workgroup (rb_conv.h) leaves
// on a loaded CU: under load such a launch waited for TWO holes on one CU (measured: a 240-workgroup launch of 60 KB
// workgroups takes ~19 us next to four streams of ResBlock launches, of 30 KB workgroups ~8 us; tools/probe/rb_diag.hip).  8 is
// enough for every k <= 5, d = 1 conv of GlowTTS (30 KB; two-way bank conflicts on half of a fragment read's lanes, in
// kernels whose LDS pipe is idle 90 % of the time); wider taps / dilations take the generic tile (host check).
#ifndef MI355TTS_G16_HALO
#define MI355TTS_G16_HALO 8
#endif
constexpr int GATE16_XW = 32 + MI355TTS_G16_HALO;  // staged columns per channel row
constexpr int GATE16_MAX_CIN = 512;  // 96 KB of LDS

struct Gate16Args {
  const float* x;  // [B][Cin][x_ld]
  long long x_bs;
  int x_ld;
  const int* len;  // valid columns per batch row: len ? len[b] * len_mul : len_const (input and output)
  int len_mul, len_const;
  const float* w;     // pack_gate16: [row tile][k-group][J][K][64 lanes]
  const float* bias;  // [row tile][16]: 8 tanh-row biases, 8 sigmoid-row biases
  int Cin, half;      // input channels, gate channels (= output rows)
  int dil, pad;
  float* y;  // [B][half][y_ld]
  long long y_bs;
  int y_ld;
  const float* cond;  // multi-speaker voices: this layer's speaker offsets of row b, [2 half], at cond + b * cond_bs (nullptr: none)
  long long cond_bs;
};

// K taps, J = 4-channel groups per k-group (Cin <= 32 J), RTW = row tiles (of 8 gate channels) per workgroup.
// RTW = 1 is the batch-1 shape (twice the workgroups of the 32-row tile: see above).  In a WIDE pass — a padded batch, a
// coalesced pass: BASELINE config 4 has ~1100 decoder columns — the 16-row tile yields ~900 workgroups that each stage the
// same [Cin x 40] tile for 60 MFMAs per wave (15.9 us per launch at 23 % of the matrix pipes): RTW = 2 stages it once for two
// row tiles.  A row tile's arithmetic does not depend on RTW (same fragments, same chains, same order of the eight partial
// sums): same bits.
template <int K, int J, int RTW = 1>
__global__ __launch_bounds__(512) void gate16_kernel(const Gate16Args a) {
  GLOW_PRIO();
  __shared__ float xs[(32 * J * GATE16_XW > 4096 * RTW) ? 32 * J * GATE16_XW : 4096 * RTW];  // [32 J][48]; afterwards the partial tiles [RTW][8][2][4][64]
  const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 6;
  const int b = blockIdx.z;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  // ragged batch: a row deals only its own tiles (conv_mfma.h, row_tiles); XCD x takes a contiguous run of them, time
  // tile fastest: an XCD's L2 then holds 1/8 of the weights (the bigger operand here) and the whole input
  const int gx = (L + 31) / 32, gy = gridDim.y;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  if (lin >= gx * gy) return;
  int tx, ty;
  {
    const int n = gx * gy, xcd = lin & 7, slot = lin >> 3, q = n >> 3, r = n & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    tx = id % gx;
    ty = id / gx;
  }
  const int t0 = tx * 32;
  constexpr int CP = 32 * J;  // staged channel rows
  constexpr int XW = GATE16_XW, XW4 = XW / 4;
  const int PA = (a.pad + 3) & ~3;
  const int nty = (a.half + 7) / 8;  // row tiles of the conv (the last workgroup row may own fewer than RTW)

  // ---- every load whose address is known at entry: the A fragments of this k-group ...
  float af[RTW][J][K];
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    const int t = ty * RTW + r < nty ? ty * RTW + r : nty - 1;
    const float* wp = a.w + ((long long)(t * 8 + kg) * (J * K)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) af[r][j][k] = wp[(j * K + k) * 64];
  }
  // ... and the activation tile (16 bytes per lane, clamped addresses, zeroed by select)
  constexpr int NF4 = CP * XW4, NE = (NF4 + 511) / 512;
  const float* xb = a.x + (long long)b * a.x_bs;
  float4 pre[NE];
  GATE_STAMP(0);
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = tid + 512 * i;
    const int row = e / XW4, f = e - row * XW4;
    const int c0 = t0 - PA + 4 * f;
    const int ci = row < a.Cin ? row : a.Cin - 1;
    pre[i] = *reinterpret_cast<const float4*>(xb + (long long)ci * a.x_ld + (c0 < 0 ? 0 : (c0 > a.x_ld - 4 ? a.x_ld - 4 : c0)));
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = tid + 512 * i;
    const int row = e / XW4, f = e - row * XW4;
    const int c0 = t0 - PA + 4 * f;
    const bool rok = row < a.Cin;
    float4 v = pre[i];
    v.x = (rok && c0 >= 0 && c0 < L) ? v.x : 0.f;
    v.y = (rok && c0 + 1 >= 0 && c0 + 1 < L) ? v.y : 0.f;
    v.z = (rok && c0 + 2 >= 0 && c0 + 2 < L) ? v.z : 0.f;
    v.w = (rok && c0 + 3 >= 0 && c0 + 3 < L) ? v.w : 0.f;
    if (e < NF4) reinterpret_cast<float4*>(xs)[e] = v;
  }
  __syncthreads();
  GATE_STAMP(1);

  // ---- main loop: B fragment lane (n = lane & 15, kq = lane >> 4) = x[4 g + kq][t0 + n + k dil - pad]
  gate_floatx4 acc0[RTW], acc1[RTW];
#pragma unroll
  for (int r = 0; r < RTW; ++r) acc0[r] = acc1[r] = gate_floatx4{0.f, 0.f, 0.f, 0.f};
  {
    const float* bp = xs + (4 * kg + (lane >> 4)) * XW + (lane & 15) + (PA - a.pad);
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float* p = bp + (32 * j) * XW + k * a.dil;
        const float b0 = p[0], b1 = p[16];
#pragma unroll
        for (int r = 0; r < RTW; ++r) {
          acc0[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j][k], b0, acc0[r], 0, 0, 0);
          acc1[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j][k], b1, acc1[r], 0, 0, 0);
        }
      }
    }
  }
  GATE_STAMP(2);
  __syncthreads();
  // ---- the k-groups' partial tiles: red[row tile][kg][column block][reg][lane]; C/D map: row = 4 (lane >> 4) + reg, col = lane & 15
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    float* red = xs + r * 4096 + (kg * 2) * 256 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[q * 64] = acc0[r][q];
      red[256 + q * 64] = acc1[r][q];
    }
  }
  __syncthreads();
  GATE_STAMP(3);
#pragma unroll
  for (int r0 = 0; r0 < RTW; r0 += 2) {
    const int r = r0 + (tid >> 8);  // threads 0-255: row tile r0, 256-511: r0 + 1
    if (r < RTW && ty * RTW + r < nty) {
      const int tt = tid & 255, tyr = ty * RTW + r;
      const int i = tt >> 5, n = tt & 31;  // gate channel 8 tyr + i, column t0 + n
      const int src = r * 4096 + (n >> 4) * 256 + (i & 3) * 64 + (i >> 2) * 16 + (n & 15);
      float v0 = a.bias[tyr * 16 + i], v1 = a.bias[tyr * 16 + 8 + i];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        v0 += xs[g * 512 + src];
        v1 += xs[g * 512 + src + 32];  // row i + 8: two 16-lane groups further
      }
      const int c = tyr * 8 + i, t = t0 + n;
      if (a.cond) {  // x_in + g_l (layers.py:154)
        const float* cd = a.cond + (long long)b * a.cond_bs + (c < a.half ? c : a.half - 1);
        v0 += cd[0];
        v1 += cd[a.half];
      }
      if (c < a.half && t < L) a.y[(long long)b * a.y_bs + (long long)c * a.y_ld + t] = tanhf(v0) * (1.0f / (1.0f + expf(-v1)));
    }
  }
  GATE_STAMP(4);
}



// ---------------------------------------------------------------------------------------------------------------------
// lin16_kernel: the same tile for the PLAIN k-tap convs of the GlowTTS encoder whose K-depth is long and whose time axis is
// short — FFN conv_1 / conv_2 (attentions.py:375-383: 192 -> 768 -> 192, k = 3), the duration predictor's convs
// (models.py:39-49) and the prenet (layers.py:73-80, k = 5).  On the generic 32-row tile these are 24-96 workgroups that walk
// their K-depth in staged chunks of 64 channels (conv_2: twelve chunks, each a global -> LDS round trip and a barrier):
// 13 us per launch at P = 120, 17 us in a padded batch of 8.  Here the whole input tile is staged ONCE, every A fragment is
// requested at entry, and 16-row tiles double the workgroup count.  Epilogue: y = act(acc + bias [+ res]).
struct Lin16Args {
  const float* x;  // [B][Cin][x_ld]
  long long x_bs;
  int x_ld;
  const int* len;  // valid columns per batch row: len ? len[b] * len_mul : len_const (input and output)
  int len_mul, len_const;
  const float* w;     // pack_lin16: [row tile][k-group][J][K][64 lanes]
  const float* bias;  // [row tile][16] (zeros when the conv has none)
  int Cin, rows;
  int dil, pad;
  float* y;  // [B][rows][y_ld]
  long long y_bs;
  int y_ld;
  const float* res;  // optional residual, geometry of y (rows < split only)
  int relu;
  // rows >= split (a multiple of 16) go to the second output, re-based: y2[row - split] (+= when accum2) — the WaveNet
  // res_skip conv's skip half (glow_tts/layers.py:154-160)
  int split;
  float* y2;
  long long y2_bs;
  int y2_ld;
  int accum2;
  // LN instantiations only: the input is LayerNorm'ed over its Cin channels per column first (glow_tts/layers.py:19-28:
  // mean / biased variance, eps inside the sqrt), optionally ReLU'ed — the producer's `norm -> [relu] -> conv` chain
  // without the LayerNorm launch.  `ln_out` (optional, geometry of x): the normalised columns of this tile are also written
  // there by the workgroups of row tile 0 (the value is needed again as a residual).
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  int ln_relu;
  float* ln_out;
};

// K taps, J = 4-channel groups per k-group (Cin <= 32 J), NBLK = 16-column blocks per workgroup (1 or 2)
// RTW = row tiles per workgroup (1: the batch-1 shape; 4: the 1 x 1 convs of WIDE passes — a padded batch's res_skip conv is
// ~900 16-row tiles that each stage the same [Cin x 40] tile for 12 MFMAs per wave; a row tile's arithmetic does not depend on
// RTW: same bits)
template <int K, int J, int NBLK, bool LN = false, int RTW = 1>
__global__ __launch_bounds__(512) void lin16_kernel(const Lin16Args a) {
  GLOW_PRIO();
  constexpr int TC = 16 * NBLK;  // columns per workgroup
  constexpr int XW = TC + MI355TTS_G16_HALO;  // staged columns per channel row
  __shared__ float xs[(32 * J * XW > 2048 * NBLK * RTW) ? 32 * J * XW : 2048 * NBLK * RTW];  // [32 J][XW]; afterwards the partial tiles [RTW][8][NBLK][4][64]
  static_assert(RTW == 1 || (TC == 32 && !LN), "several row tiles per workgroup: the 512 threads are one row tile's epilogue");
  // two workgroups per CU (160 KB of LDS): every form stays under 80 KB, and the RTW = 4 form — whose reduction scratch, not its
  // staged tile, sets the size: exactly 64 KB — under that; a wider HALO / NBLK / RTW must not pass either silently
  // (tests/test_kernel_resources.py bounds the compiled figures too)
  static_assert(sizeof(float) * ((32 * J * XW > 2048 * NBLK * RTW) ? 32 * J * XW : 2048 * NBLK * RTW) <= (RTW > 1 ? 64 : 80) * 1024,
                "lin16 tile: more LDS than two workgroups per CU allow");
  __shared__ float lnred[LN ? 8 * 64 : 1];
  static_assert(!LN || XW <= 64, "LayerNorm prologue: one lane per staged column");
  const int tid = threadIdx.x, lane = tid & 63, kg = tid >> 6;
  const int b = blockIdx.z;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int gx = (L + TC - 1) / TC, gy = gridDim.y;
  const int lin = blockIdx.x + blockIdx.y * gridDim.x;
  if (lin >= gx * gy) return;  // ragged batch: a row deals only its own tiles
  int tx, ty;
  {
    const int n = gx * gy, xcd = lin & 7, slot = lin >> 3, q = n >> 3, r = n & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    tx = id % gx;
    ty = id / gx;
  }
  const int t0 = tx * TC;
  constexpr int CP = 32 * J;
  constexpr int XW4 = XW / 4;
  const int PA = (a.pad + 3) & ~3;

  // ---- every load whose address is known at entry
  const int nty = (a.rows + 15) / 16;
  float af[RTW][J][K];
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    const int t = ty * RTW + r < nty ? ty * RTW + r : nty - 1;
    const float* wp = a.w + ((long long)(t * 8 + kg) * (J * K)) * 64 + lane;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int k = 0; k < K; ++k) af[r][j][k] = wp[(j * K + k) * 64];
  }
  constexpr int NF4 = CP * XW4, NE = (NF4 + 511) / 512;
  const float* xb = a.x + (long long)b * a.x_bs;
  float4 pre[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = tid + 512 * i;
    const int row = e / XW4, f = e - row * XW4;
    const int c0 = t0 - PA + 4 * f;
    const int ci = row < a.Cin ? row : a.Cin - 1;
    pre[i] = *reinterpret_cast<const float4*>(xb + (long long)ci * a.x_ld + (c0 < 0 ? 0 : (c0 > a.x_ld - 4 ? a.x_ld - 4 : c0)));
  }
  // the epilogue's operands too: thread (row i, column n) of the tile
  const int ei = tid / TC, en = tid - ei * TC;
  const bool ethread = tid < 16 * TC;
  const int et = t0 + en;
  int erow[RTW];
  bool eok[RTW], second[RTW];
  float ebias[RTW], eres[RTW];
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    const int tyr = ty * RTW + r < nty ? ty * RTW + r : nty - 1;
    erow[r] = tyr * 16 + (ethread ? ei : 0);
    eok[r] = ethread && ty * RTW + r < nty && erow[r] < a.rows && et < L;
    ebias[r] = a.bias[tyr * 16 + (ethread ? ei : 0)];
    second[r] = tyr * 16 >= a.split;  // uniform per row tile
    eres[r] = 0.f;
    if (!second[r]) {
      if (a.res) eres[r] = a.res[(long long)b * a.y_bs + (long long)(eok[r] ? erow[r] : 0) * a.y_ld + (eok[r] ? et : 0)];
    } else if (a.accum2) {
      eres[r] = a.y2[(long long)b * a.y2_bs + (long long)(eok[r] ? erow[r] - a.split : 0) * a.y2_ld + (eok[r] ? et : 0)];
    }
  }
  float lg[LN ? 4 * J : 1], lb[LN ? 4 * J : 1];  // LayerNorm gamma / beta of this wave's rows (kg, kg + 8, ...)
  if constexpr (LN) {
#pragma unroll
    for (int i = 0; i < 4 * J; ++i) {
      const int r = kg + 8 * i;
      lg[i] = a.ln_gamma[r < a.Cin ? r : a.Cin - 1];
      lb[i] = a.ln_beta[r < a.Cin ? r : a.Cin - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = tid + 512 * i;
    const int row = e / XW4, f = e - row * XW4;
    const int c0 = t0 - PA + 4 * f;
    const bool rok = row < a.Cin;
    float4 v = pre[i];
    v.x = (rok && c0 >= 0 && c0 < L) ? v.x : 0.f;
    v.y = (rok && c0 + 1 >= 0 && c0 + 1 < L) ? v.y : 0.f;
    v.z = (rok && c0 + 2 >= 0 && c0 + 2 < L) ? v.z : 0.f;
    v.w = (rok && c0 + 3 >= 0 && c0 + 3 < L) ? v.w : 0.f;
    if (e < NF4) reinterpret_cast<float4*>(xs)[e] = v;
  }
  __syncthreads();
  if constexpr (LN) {
    // ---- LayerNorm of the staged tile, column by column (halo columns included: the conv reads them; columns outside
    // [0, L) stay zero = the conv's padding).  Lane = staged column, wave g = rows g, g + 8, ... (4 J of them, in registers)
    constexpr int RW = 4 * J;
    const int j = lane < XW ? lane : XW - 1, g = kg;
    const int col = t0 - PA + j;
    const bool cok = lane < XW && col >= 0 && col < L;
    float v[RW];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int r = g + 8 * i;
      v[i] = r < a.Cin ? xs[r * XW + j] : 0.f;
      sum += v[i];
    }
    lnred[g * 64 + lane] = sum;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) mean += lnred[q * 64 + lane];
    mean /= (float)a.Cin;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const float d = (g + 8 * i < a.Cin) ? v[i] - mean : 0.f;
      sq += d * d;
    }
    lnred[g * 64 + lane] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) var += lnred[q * 64 + lane];
    var /= (float)a.Cin;
    const float rstd = rsqrtf(var + a.ln_eps);
    const bool wout = a.ln_out != nullptr && ty == 0 && cok && j >= PA && j < PA + TC;  // this tile's own columns
    float* ob = a.ln_out ? a.ln_out + (long long)b * a.x_bs + (cok ? col : 0) : nullptr;
    if (cok) {
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int r = g + 8 * i;
        if (r < a.Cin) {
          float o = (v[i] - mean) * rstd * lg[i] + lb[i];
          if (a.ln_relu) o = fmaxf(o, 0.f);
          xs[r * XW + j] = o;
          if (wout) ob[(long long)r * a.x_ld] = o;
        }
      }
    }
    __syncthreads();
  }

  // ---- main loop: B fragment lane (n = lane & 15, kq = lane >> 4) = x[4 (g + 8 j) + kq][t0 + n + k dil - pad]
  gate_floatx4 acc[RTW][NBLK];
#pragma unroll
  for (int r = 0; r < RTW; ++r)
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) acc[r][nb] = gate_floatx4{0.f, 0.f, 0.f, 0.f};
  {
    const float* bp = xs + (4 * kg + (lane >> 4)) * XW + (lane & 15) + (PA - a.pad);
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float* p = bp + (32 * j) * XW + k * a.dil;
        float bf[NBLK];
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) bf[nb] = p[16 * nb];
#pragma unroll
        for (int r = 0; r < RTW; ++r)
#pragma unroll
          for (int nb = 0; nb < NBLK; ++nb) acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][j][k], bf[nb], acc[r][nb], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  // ---- the k-groups' partial tiles: red[row tile][kg][column block][reg][lane]; C/D map: row = 4 (lane >> 4) + reg, col = lane & 15
#pragma unroll
  for (int r = 0; r < RTW; ++r) {
    float* red = xs + r * (2048 * NBLK) + (kg * NBLK) * 256 + lane;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[nb * 256 + q * 64] = acc[r][nb][q];
  }
  __syncthreads();
  if (ethread) {
    const int src = (en >> 4) * 256 + (ei & 3) * 64 + (ei >> 2) * 16 + (en & 15);
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) v += xs[r * (2048 * NBLK) + g * (NBLK * 256) + src];
      v += ebias[r];
      v += eres[r];
      if (a.relu && !second[r]) v = v > 0.f ? v : 0.f;
      if (eok[r]) {
        if (!second[r]) a.y[(long long)b * a.y_bs + (long long)erow[r] * a.y_ld + et] = v;
        else a.y2[(long long)b * a.y2_bs + (long long)(erow[r] - a.split) * a.y2_ld + et] = v;
      }
    }
  }
}

}  // namespace mi355tts
