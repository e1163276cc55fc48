// One WaveNet layer of a GlowTTS coupling block per launch, on column-owner workgroups: the THROUGHPUT form of the decoder
// (glow_tts/layers.py:138-162 — x_in = in_layers[i](x) [+ g_l]; acts = tanh(x_in[:H]) * sigmoid(x_in[H:])
// (glow_tts/utils.py:31-38); res_skip = res_skip_layers[i](acts); x = (x + res_skip[:H]) * mask; output += res_skip[H:]).
//
// STATUS (round 6): NOT part of the product library any more — an experiment record.  It was compiled into libmi355tts.so
// behind option "wn_layer" (default 0) in round 5 (commit 4c32bea and before: csrc/wn_layer.h, the dispatch in glow_forward.h /
// host_launch.h, tests/test_emu_wn_layer.py), measured slower under every load and taken out; nothing includes this file.  The form was built on the reading that, next to other
// calls, a GlowTTS launch costs the CU residency it takes from the vocoder's ResBlock workgroups, not its latency (VERDICT
// r04): gate16.h / lin16_kernel run this layer as 240 + 120 workgroups of 512 threads that each stage a [H x 40] tile, run
// 30-60 MFMAs per wave and meet in LDS (8 + 5 us on an idle chip at 3-10 % of the matrix pipes); here a workgroup OWNS 16 time
// columns with all their channels and runs the layer's 442 k MAC per column out of LDS — 20 workgroups for a 312-column
// decoder, the activations of a layer never leave the CU, gate conv + gate + res_skip are one launch.  On the device
// (profiles/r05_wn_layer_ab.txt): 40.6 us per layer alone (1.77 MB of fragments through one CU with 30 KB in flight), 72.5 us
// under load; 243 utterances/s against the chain's 281 with eight calls in flight, BASELINE config 4 2.48 ms per call
// against 2.00 (a padded batch of 8 is 70 column owners on 256 CUs).  Under load many short workgroups that request all
// their operands at entry beat few long ones; the kernel stays as the tested, bit-identical record of that answer.
//
// Shape of a workgroup: 4 waves (one per SIMD), <= 128 VGPRs, 31 KB of LDS — the hole ONE finishing ResBlock workgroup
// (rb_conv.h: 4 waves, 32 KB) leaves on a loaded CU, so a launch needs no drained CU to start.
//
// Arithmetic: EXACTLY the lone-call kernels' — same packed fragments (pack_gate16 / pack_lin16: [row tile][k-group g]
// [J][K][64 lanes]), same v_mfma_f32_16x16x4_f32 chains (partial g = the 4-channel groups g, g + 8, ... with all their
// taps, in that order), and the eight partials folded in the order gate16_kernel / lin16_kernel sum them in LDS
// (bias + p0 + ... + p7; 0 + p0 + ... + p7 + bias + residual) — so a call computes the same BITS whichever form the
// host picks, and the choice may depend on the load (tests: test_emu_wn_layer.py, test_gpu_parity.py).
//   wave w of 4 takes the row tiles {(p RT + r) 4 + w : r < RT} in pass p: RT = 3 tiles share every B fragment (one
//   ds_read_b32 per 3 MFMAs); a tile's fragment stream is contiguous (8 J K steps of 64 dwords) and runs through a
//   register ring 2 K steps deep that never drains: the last steps of a pass request the first fragments of the next
//   pass (or of the res_skip phase).  Partials alternate between two accumulator sets, the finished one is folded into
//   the running sum a few steps into the next chain.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "prio.h"

namespace mi355tts {

typedef float wn_floatx4 __attribute__((ext_vector_type(4)));

constexpr int WN_T = 16;                             // columns per workgroup
constexpr int WN_XW = WN_T + MI355TTS_G16_HALO;      // staged columns per channel row
constexpr int WN_RT = 3;                             // row tiles per wave and pass

struct WnLayerArgs {
  const float* x;   // [B][H][ld]: the layer's input (the WaveNet's hidden state)
  float* x_out;     // [B][H][ld]: x + res_skip[:H] — NOT x (a neighbour's halo); unused by the gate-only form
  long long bs;
  int ld;
  const int* len;  // valid columns per batch row: len ? len[b] * len_mul : len_const
  int len_mul, len_const;
  const float* gw;  // pack_gate16 of in_layers[i]
  const float* gb;
  const float* rw;  // pack_lin16 of res_skip_layers[i] (2H rows); unused by the gate-only form
  const float* rb;
  float* acts;  // gate-only form (the block's last layer: its res_skip lives in glow_tail_kernel): [B][H][ld]
  float* skip;  // [B][H][ld]: = res_skip[H:] (accum 0) or += (accum 1)
  int accum;
  int pad;  // (K - 1) / 2, dilation 1
  const float* cond;  // multi-speaker voices: this layer's speaker offsets of row b, [2H], at cond + b * cond_bs
  long long cond_bs;
};

template <int N, class F, int I = 0>
__device__ __forceinline__ void wn_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    wn_static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// K taps, J = 4-channel groups per k-group (H = 32 J input = gate channels), RES = with the res_skip conv
template <int K, int J, bool RES>
__global__ __launch_bounds__(256, 4) void wn_layer_kernel(const WnLayerArgs a) {
  GLOW_PRIO();
  constexpr int H = 32 * J, RT = WN_RT, NP = J / RT, D = 2 * K;
  constexpr int GS = J * K;    // gate steps per partial (k-group)
  constexpr int TS = 8 * GS;   // ... per row tile
  constexpr int RS = 8 * J;    // res_skip steps per row tile
  constexpr int XW = WN_XW, XW4 = XW / 4;
  static_assert(J % RT == 0 && NP % 2 == 0, "row tiles: whole passes, first half of the res_skip passes = residual rows");
  static_assert((2 * GS) % D == 0 && (2 * GS) % 3 == 0 && K >= 2, "ring slots are static");
  __shared__ float xs[H * XW];    // lrelu-free input tile [H][XW], zero outside [0, L)
  __shared__ float as[H * WN_T];  // gated activations [H][16] (B operand of the res_skip conv)
  // the gate conv's biases: a running sum starts from them inside the loop of trips, and a global load consumed there
  // makes the compiler's s_waitcnt drain the fragment ring at every trip (vmcnt counts in order) — LDS reads do not
  __shared__ float gbs[H * 2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int t0 = blockIdx.x * WN_T;
  if (t0 >= L) return;  // ragged batch: a row owns only its own column tiles
  const int PA = (a.pad + 3) & ~3;
  const int n = lane & 15, q = lane >> 4;

  // ---- every load whose address is known at entry: the first ring of fragments ...
  // (wave-uniform tile pointers + the lane as a 32-bit offset: the loads take their base from SGPRs)
  auto gtile = [&](int p, int r) { return a.gw + (long long)(((p * RT + r) * 4 + w) * TS) * 64; };
  auto rtile = [&](int p, int r) { return a.rw + (long long)(((p * RT + r) * 4 + w) * RS) * 64; };
  float ring[RT][D];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const float* p0 = gtile(0, r);
#pragma unroll
    for (int i = 0; i < D; ++i) ring[r][i] = p0[i * 64 + lane];
  }
  // ... and the activation tile (16 bytes per lane, clamped addresses, zeroed by select)
  constexpr int NF4 = H * XW4, NE = (NF4 + 255) / 256;
  const float* xb = a.x + (long long)b * a.bs;
  {
    float4 pre[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      const int row = e / XW4 < H ? e / XW4 : H - 1, f = e - (e / XW4) * XW4;
      const int c0 = t0 - PA + 4 * f;
      pre[i] = *reinterpret_cast<const float4*>(xb + (long long)row * a.ld + (c0 < 0 ? 0 : (c0 > a.ld - 4 ? a.ld - 4 : c0)));
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      const int f = e - (e / XW4) * XW4;
      const int c0 = t0 - PA + 4 * f;
      float4 v = pre[i];
      v.x = (c0 >= 0 && c0 < L) ? v.x : 0.f;
      v.y = (c0 + 1 >= 0 && c0 + 1 < L) ? v.y : 0.f;
      v.z = (c0 + 2 >= 0 && c0 + 2 < L) ? v.z : 0.f;
      v.w = (c0 + 3 >= 0 && c0 + 3 < L) ? v.w : 0.f;
      if (e < NF4) reinterpret_cast<float4*>(xs)[e] = v;
    }
  }
  if (tid < H / 2) reinterpret_cast<float4*>(gbs)[tid] = reinterpret_cast<const float4*>(a.gb)[tid];
  __syncthreads();

  const int t = t0 + n;
  const bool tok = t < L;
  // ---- gate conv: B fragment of step (g, j, k), lane (n, q) = x[4 (g + 8 j) + q][t0 + n + k - pad]
  const float* bptr = xs + q * XW + n + (PA - a.pad);
  wn_static_for<NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    wn_floatx4 sum[RT], part[2][RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int T = (p * RT + r) * 4 + w;
      const float4 bv = *reinterpret_cast<const float4*>(gbs + T * 16 + 4 * q);
      sum[r] = wn_floatx4{bv.x, bv.y, bv.z, bv.w};
      part[0][r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
      part[1][r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const float* nxt[RT];  // where the fragment stream continues behind this pass
#pragma unroll
    for (int r = 0; r < RT; ++r) nxt[r] = p + 1 < NP ? gtile(p + 1 < NP ? p + 1 : p, r) : (RES ? rtile(0, r) : gtile(p, r));
    float bq[3];  // B fragments run two steps ahead of their MFMAs
    bq[0] = bptr[0];
    bq[1] = bptr[1];
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {  // two partials (k-groups 2 it, 2 it + 1) per trip
      const float* bg = bptr + it * (8 * XW);
      const float* lo[RT];
      const float* hi[RT];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        lo[r] = gtile(p, r) + (long long)it * (2 * GS * 64);
        hi[r] = it < 3 ? lo[r] + 2 * GS * 64 : nxt[r];
      }
      wn_static_for<2 * GS>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int s = u / GS, i = u % GS, slot = u % D;
        // the B fragment two steps on (the first ones of the next trip behind the last ones: rows 8 XW further on)
        constexpr int u2 = u + 2, s2 = u2 / GS, i2 = u2 % GS;
        bq[u2 % 3] = bg[(4 * s2 + 32 * (i2 / K)) * XW + (i2 % K)];
#pragma unroll
        for (int r = 0; r < RT; ++r) part[s][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[r][slot], bq[u % 3], part[s][r], 0, 0, 0);
        // refill this slot with the step D further on
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          if constexpr (u + D < 2 * GS) ring[r][slot] = lo[r][(u + D) * 64 + lane];
          else ring[r][slot] = hi[r][(u + D - 2 * GS) * 64 + lane];
        }
        // the other set's finished partial joins the running sum a few steps into this chain
        if constexpr (i == 2) {
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            sum[r] += part[1 - s][r];
            part[1 - s][r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      });
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) sum[r] += part[1][r];
    // ---- gate: C/D map row = 4 q + reg: tanh rows (0-7) in lanes 0-31, their sigmoid rows (8-15) 32 lanes further on
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int T = (p * RT + r) * 4 + w;
      wn_floatx4 v = sum[r];
      if (a.cond) {  // x_in + g_l (layers.py:154)
        const float4 cv = *reinterpret_cast<const float4*>(a.cond + (long long)b * a.cond_bs + (q < 2 ? 0 : H - 8) + 8 * T + 4 * q);
        v[0] += cv.x;
        v[1] += cv.y;
        v[2] += cv.z;
        v[3] += cv.w;
      }
      wn_floatx4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = __shfl_xor(v[e], 32);
      // lanes 0-31 finish registers 0 and 1 of their rows, lanes 32-63 registers 2 and 3 of the partner's
      const bool up = lane >= 32;
      const float ta = up ? o[2] : v[0], sa = up ? v[2] : o[0];
      const float tb = up ? o[3] : v[1], sb = up ? v[3] : o[1];
      const float ga = tanhf(ta) * (1.0f / (1.0f + expf(-sa)));
      const float gb2 = tanhf(tb) * (1.0f / (1.0f + expf(-sb)));
      const int c = 8 * T + 4 * (q & 1) + (up ? 2 : 0);
      if constexpr (RES) {
        as[c * WN_T + n] = tok ? ga : 0.f;
        as[(c + 1) * WN_T + n] = tok ? gb2 : 0.f;
      } else if (tok) {
        float* yp = a.acts + (long long)b * a.bs + (long long)c * a.ld + t;
        yp[0] = ga;
        yp[a.ld] = gb2;
      }
    }
  });

  if constexpr (RES) {
    // ---- res_skip conv (1 x 1, H -> 2H): B fragment of step (g, j), lane (n, q) = acts[4 (g + 8 j) + q][n].  Same trips as the
    // gate conv (two partials of J steps each), the ring 2 J steps deep: the gate conv's last trip filled its first D slots
    constexpr int RD = 2 * J;
    static_assert(RD >= D && RD % 3 == 0, "the ring handed over by the gate phase fits the res_skip ring");
    float rr[RT][RD];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        if (i < D) rr[r][i] = ring[r][i];
        else rr[r][i] = rtile(0, r)[i * 64 + lane];
      }
    }
    __syncthreads();
    wn_static_for<NP>([&](auto pc) {
      constexpr int p = decltype(pc)::value;
      constexpr bool second = p >= NP / 2;  // the skip half
      wn_floatx4 sum[RT], part[2][RT];
      wn_floatx4 old[RT];  // what the epilogue adds: x (from the staged tile) or the skip sum so far
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        sum[r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
        part[0][r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
        part[1][r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
        old[r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
        if constexpr (second) {
          if (a.accum) {
            const int row = 16 * ((p * RT + r) * 4 + w) + 4 * q;
            const float* sp = a.skip + (long long)b * a.bs + (long long)(row - H) * a.ld + (tok ? t : 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) old[r][e] = sp[(long long)e * a.ld];
          }
        }
      }
      const float* ap = as + lane;
      float bq[3];
      bq[0] = ap[0];
      bq[1] = ap[(J > 1 ? 32 : 4) * WN_T];
#pragma unroll 1
      for (int it = 0; it < 4; ++it) {
        const float* ag = ap + it * (8 * WN_T);
        const float* hi[RT];  // the ring is a whole trip deep: every refill is the same step of the next trip
#pragma unroll
        for (int r = 0; r < RT; ++r) hi[r] = it < 3 ? rtile(p, r) + (it + 1) * (RD * 64) : rtile(p + 1 < NP ? p + 1 : p, r);
        wn_static_for<RD>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          constexpr int s = u / J;
          constexpr int u2 = u + 2, s2 = u2 / J, j2 = u2 % J;
          bq[u2 % 3] = ag[(4 * s2 + 32 * j2) * WN_T];
#pragma unroll
          for (int r = 0; r < RT; ++r) part[s][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(rr[r][u], bq[u % 3], part[s][r], 0, 0, 0);
#pragma unroll
          for (int r = 0; r < RT; ++r) rr[r][u] = hi[r][u * 64 + lane];
          if constexpr (u % J == 2) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
              sum[r] += part[1 - s][r];
              part[1 - s][r] = wn_floatx4{0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
      }
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        sum[r] += part[1][r];
        const int T = (p * RT + r) * 4 + w;
        const int row = 16 * T + 4 * q;
        const float4 bv = *reinterpret_cast<const float4*>(a.rb + T * 16 + 4 * q);
        if constexpr (!second) {
#pragma unroll
          for (int e = 0; e < 4; ++e) old[r][e] = xs[(row + e) * XW + PA + n];
        }
        wn_floatx4 v = sum[r];
        v[0] += bv.x;
        v[1] += bv.y;
        v[2] += bv.z;
        v[3] += bv.w;
        v += old[r];
        if (tok) {
          float* yp = (second ? a.skip + (long long)(row - H) * a.ld : a.x_out + (long long)row * a.ld) + (long long)b * a.bs + t;
#pragma unroll
          for (int e = 0; e < 4; ++e) yp[(long long)e * a.ld] = v[e];
        }
      }
    });
  }
}

}  // namespace mi355tts
