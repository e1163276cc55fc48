// Column-owner launches of the GlowTTS path: chains of small dense contractions whose seams are channel-mixing steps
// (every output row of a column needs ALL rows of the previous step's column), fused into ONE launch by letting a
// workgroup own 16 time columns with all their channels.
//
//   glow_tail_kernel  — the end of a coupling block and the start of the next one (glow_tts/attentions.py:119-142,
//                       layers.py:138-162, :192-194, :238-272), five launches' worth of the reverse flow in one:
//                         s      = skip + res_skip_layers[last](acts)          (1 x 1, H -> H; the last layer is all skip)
//                         m|logs = end(s)                                      (1 x 1, H -> 2 half)
//                         z1     = (z1 - m) * exp(-logs)
//                         z      = ActNorm^-1(InvConvNear^-1(z))               (n_split = 4 channel groups)
//                         h      = start_next(z0)                              (1 x 1, half -> H; next block in reverse order)
//   oproj_ln_kernel   — conv_o of the encoder's attention + residual + LayerNorm (attentions.py:62-68, :205-212)
//
// Why column owners here and not in the WaveNet layers: these steps are 15-37 k MAC per column — a 16-column tile is
// 1296 v_mfma_f32_16x16x4_f32 over four SIMDs (~5 us of matrix-pipe time at one workgroup per CU; the launch measures
// 12 us, profiles/NOTES.md) against 3 launches of 5.9-7.4 us each, which are launch-latency chains (cold weights -> MFMA
// -> k-group reduction -> store) on a ~312-column problem.  The gate convs (442 k MAC per column) would take 26 us on a
// column owner and stay row-tiled (gate16.h).
//
// GEMM form: Y[R x 16] = W[R x K] X[K x 16].  A fragments pre-packed per lane (pack_col16: one float4 = four k-steps),
// streamed from L2 through a register ring; X lives in LDS as [K][16] — the B fragment of k-step s is the 64
// consecutive floats at 64 s (lane = 16 kq + n), conflict-free; C/D map: row = 4 (lane >> 4) + reg, col = lane & 15.
// Wave v of 8 takes row tiles v and v + 8: waves v and v + 4 share a SIMD, so every SIMD's matrix pipe carries three
// of the twelve row tiles of H = 192.
#pragma once
#include <hip/hip_runtime.h>
#include "prio.h"
#include <type_traits>

namespace mi355tts {

#ifndef COL_STAMP
#define COL_STAMP(n)  // phase stamps of tools/probe/coltile_bench.hip
#endif

typedef float col_floatx4 __attribute__((ext_vector_type(4)));

constexpr int COL_T = 16;        // columns per workgroup
constexpr int COL_MAXROWS = 256;  // rows / K-depth of any step (two row tiles per wave)
constexpr int COL_RING = 6;       // float4 A fragments in flight per row tile (24 k-steps)

// One step of a chain for this wave: row tiles `wave` and `wave + 8` of `RT` (nt = how many of them exist).  The A ring is
// filled and refilled UNCONDITIONALLY for both tiles from clamped addresses (a missing tile re-reads the other one's
// fragments: L1 hits) — a load behind a branch makes the compiler drain vmcnt(0) at every join, and ring arrays that
// are written on some paths only end up in scratch; only the MFMAs are under (wave-uniform) conditions.
struct ColStep {
  const float4* w;  // [RT][KQ4][64]
  int RT, KQ4;
};
__device__ __forceinline__ int col_tiles_of(int wave, int RT) { return wave + 8 < RT ? 2 : (wave < RT ? 1 : 0); }

// ring prologue: the first COL_RING fragments.  Called for the NEXT step as soon as the current step's MFMAs are issued,
// i.e. before the epilogue and the barrier the chain needs anyway.
__device__ __forceinline__ void col_step_fill(const ColStep& st, int wave, float4 (&a0)[COL_RING], float4 (&a1)[COL_RING]) {
  const int lane = threadIdx.x & 63;
  const int r0 = wave < st.RT ? wave : 0, r1 = wave + 8 < st.RT ? wave + 8 : r0;
  const float4* p0 = st.w + (long long)r0 * st.KQ4 * 64 + lane;
  const float4* p1 = st.w + (long long)r1 * st.KQ4 * 64 + lane;
#pragma unroll
  for (int j = 0; j < COL_RING; ++j) {
    const int q = j < st.KQ4 ? j : st.KQ4 - 1;
    a0[j] = p0[q * 64];
    a1[j] = p1[q * 64];
  }
}

template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// the four B fragments of k-step group q: X = LDS [16 KQ4][16], 64 consecutive floats per k-step
__device__ __forceinline__ void col_b_read(const float* xl, int q, float (&bq)[4]) {
  const float* xb = xl + q * 256;
  bq[0] = xb[0];
  bq[1] = xb[64];
  bq[2] = xb[128];
  bq[3] = xb[192];
}
template <int NT>
__device__ __forceinline__ void col_mfma4(const float4& f0, const float4& f1, const float (&bq)[4], col_floatx4& acc0, col_floatx4& acc1) {
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.x, bq[0], acc0, 0, 0, 0);
  if constexpr (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.x, bq[0], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.y, bq[1], acc0, 0, 0, 0);
  if constexpr (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.y, bq[1], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.z, bq[2], acc0, 0, 0, 0);
  if constexpr (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.z, bq[2], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f0.w, bq[3], acc0, 0, 0, 0);
  if constexpr (NT == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.w, bq[3], acc1, 0, 0, 0);
}
// Main loop over KQ4 groups of four k-steps for NT row tiles.  Whole chunks of COL_RING groups run branch-free (B
// fragments read one group ahead, ring slots refilled right after their use); only the remainder (KQ4 % COL_RING groups)
// is under wave-uniform guards.  (A guard per group in the main body made the compiler copy the accumulators at every
// join behind an s_nop drain of the matrix pipe, and kept the LDS reads from moving ahead of the MFMAs.)
template <int NT>
__device__ __forceinline__ void col_gemm(const float4* p0, const float4* p1, int KQ4, const float* X, float4 (&a0)[COL_RING],
                                         float4 (&a1)[COL_RING], col_floatx4& acc0, col_floatx4& acc1) {
  const float* xl = X + (threadIdx.x & 63);
  const int last = KQ4 - 1;
  float bc[4], bn[4];
  col_b_read(xl, 0, bc);
  int q0 = 0;
  for (; q0 + COL_RING <= KQ4; q0 += COL_RING) {
    static_for<COL_RING>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int q = q0 + j;
      col_b_read(xl, q + 1 < KQ4 ? q + 1 : last, bn);
      col_mfma4<NT>(a0[j], a1[j], bc, acc0, acc1);
      // refill the slot of the PREVIOUS group (its registers are free; this group's are being read by its MFMAs)
      constexpr int jp = (j + COL_RING - 1) % COL_RING;
      const int qn = q - 1 + COL_RING < KQ4 ? q - 1 + COL_RING : last;
      a0[jp] = p0[qn * 64];
      if constexpr (NT == 2) a1[jp] = p1[qn * 64];
#pragma unroll
      for (int i = 0; i < 4; ++i) bc[i] = bn[i];
      // issue order inside the group (0x008 = MFMA, 0x020 = VMEM read, 0x100 = LDS read): the ring refills and the next
      // group's B reads go out behind this group's first MFMAs — left alone, the scheduler sinks all refills of a chunk
      // to its end, which shortens the ring to nothing at every chunk boundary
#pragma unroll
      for (int i = 0; i < 4 * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < NT) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        else if (i < NT + 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    });
  }
  const int rem = KQ4 - q0;
#pragma unroll
  for (int j = 0; j < COL_RING - 1; ++j)
    if (j < rem) {
      col_b_read(xl, q0 + j + 1 < KQ4 ? q0 + j + 1 : last, bn);
      col_mfma4<NT>(a0[j], a1[j], bc, acc0, acc1);
#pragma unroll
      for (int i = 0; i < 4; ++i) bc[i] = bn[i];
    }
}
__device__ __forceinline__ void col_step_run(const ColStep& st, int wave, const float* X, float4 (&a0)[COL_RING], float4 (&a1)[COL_RING],
                                             col_floatx4& acc0, col_floatx4& acc1) {
  const int lane = threadIdx.x & 63;
  const int nt = col_tiles_of(wave, st.RT);
  const int r0 = wave < st.RT ? wave : 0, r1 = wave + 8 < st.RT ? wave + 8 : r0;
  const float4* p0 = st.w + (long long)r0 * st.KQ4 * 64 + lane;
  const float4* p1 = st.w + (long long)r1 * st.KQ4 * 64 + lane;
  acc0 = col_floatx4{0.f, 0.f, 0.f, 0.f};
  acc1 = col_floatx4{0.f, 0.f, 0.f, 0.f};
  if (nt == 2) col_gemm<2>(p0, p1, st.KQ4, X, a0, a1, acc0, acc1);
  else if (nt == 1) col_gemm<1>(p0, p1, st.KQ4, X, a0, a1, acc0, acc1);
}

// A [rows][16] tile of a [B][rows][ld] tensor goes to LDS in two halves so that EVERY global load of a launch is in flight
// before the first one is waited for (these launches are latency chains: a load -> store loop with a run-time trip
// count costs one memory round trip per iteration): col_tile_issue requests the (up to two) float4s a thread owns from
// clamped addresses, col_tile_land masks them (rows >= rows, columns >= L -> 0) and writes rows [0, rows_pad).
struct ColTileRegs {
  float4 v[2];
};
static_assert(COL_MAXROWS * 4 <= 2 * 512, "two float4 per thread cover a tile");
__device__ __forceinline__ void col_tile_issue(const float* src, int rows, int ld, int t0, ColTileRegs& r) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = threadIdx.x + 512 * i;
    const int row = e >> 2, f = e & 3;
    const int c0 = t0 + 4 * f;
    const int rc = row < rows ? row : rows - 1;
    const int cc = c0 <= ld - 4 ? c0 : ld - 4;
    r.v[i] = *reinterpret_cast<const float4*>(src + (long long)rc * ld + cc);
  }
}
__device__ __forceinline__ void col_tile_land(const ColTileRegs& r, bool present, int rows, int rows_pad, int t0, int L, float* dst) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = threadIdx.x + 512 * i;
    const int row = e >> 2, f = e & 3;
    const int c0 = t0 + 4 * f;
    const bool rok = present && row < rows;
    float4 v = r.v[i];
    v.x = (rok && c0 < L) ? v.x : 0.f;
    v.y = (rok && c0 + 1 < L) ? v.y : 0.f;
    v.z = (rok && c0 + 2 < L) ? v.z : 0.f;
    v.w = (rok && c0 + 3 < L) ? v.w : 0.f;
    if (row < rows_pad) reinterpret_cast<float4*>(dst)[e] = v;
  }
}
// the biases of this wave's (up to two) row tiles in C/D order: bb[i][r] = bias[(wave + 8 i) 16 + 4 (lane >> 4) + r]; the
// packed bias covers whole row tiles, a missing tile re-reads the first one's
__device__ __forceinline__ void col_bias_issue(const float* bias, int RT, int wave, int rq, float (&bb)[2][4]) {
  const int r0 = wave < RT ? wave : 0, r1 = wave + 8 < RT ? wave + 8 : r0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    bb[0][r] = bias[r0 * 16 + rq + r];
    bb[1][r] = bias[r1 * 16 + rq + r];
  }
}

struct GlowTailArgs {
  const float* acts;  // [B][H][h_ld]: gate output of the last WaveNet layer
  const float* skip;  // [B][H][h_ld]: skip sum of the earlier layers (nullptr when the block has one layer)
  float* hnext;       // [B][H][h_ld]: start(z0) of the next block in reverse order (nullptr = last block)
  long long h_bs;
  int h_ld;
  float* z;  // [B][2 half][z_ld]: the flow tensor, updated in place
  long long z_bs;
  int z_ld;
  const int* len;  // valid columns of row b: len ? len[b] * len_mul : len_const
  int len_mul, len_const;
  const float *w_rs, *b_rs;    // pack_col16 of res_skip_layers[last]: H rows, K = H
  const float *w_end, *b_end;  // end: 2 half rows (m rows first, then logs), K = H
  const float *w_st, *b_st;    // next block's start: H rows, K = half (the block's own start when hnext is null: unused)
  const float* mix_w;          // [4][4] pre-inverted InvConvNear weight
  const float* mix_bias;       // [2 half] ActNorm bias
  const float* mix_scale;      // [2 half] exp(-logs) of ActNorm
  int H, half;
};

__global__ __launch_bounds__(512) void glow_tail_kernel(const GlowTailArgs a) {
  GLOW_PRIO();
  // R1: acts, then m|logs;  R2: skip -> s, then the new z0;  R3: z
  __shared__ float lds[3 * COL_MAXROWS * COL_T];
  float* R1 = lds;
  float* R2 = lds + COL_MAXROWS * COL_T;
  float* R3 = lds + 2 * COL_MAXROWS * COL_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int t0 = blockIdx.x * COL_T;
  if (t0 >= L) return;
  const int H = a.H, half = a.half, C = 2 * half;
  const int KH = (H + 15) & ~15, KZ = (half + 15) & ~15, CP = (C + 15) & ~15;
  const ColStep s_rs{reinterpret_cast<const float4*>(a.w_rs), KH / 16, KH / 16};
  const ColStep s_end{reinterpret_cast<const float4*>(a.w_end), CP / 16, KH / 16};
  const ColStep s_st{reinterpret_cast<const float4*>(a.w_st), KH / 16, KZ / 16};
  const int col = lane & 15, rq = 4 * (lane >> 4);

  // ---- every load whose address is known at entry, in ONE batch: the first weight fragments (cold), the three
  // activation tiles, the biases of all three steps, the ActNorm constants of this thread's channel groups
  // (the tiles first: the barrier waits for them, the 96 KB of ring fragments — 1.5 k cycles of this CU's L1 path — only
  // have to be there when the first MFMAs issue)
  float4 a0[COL_RING], a1[COL_RING];
  COL_STAMP(0);
  ColTileRegs t_acts, t_skip, t_z;
  col_tile_issue(a.acts + (long long)b * a.h_bs, H, a.h_ld, t0, t_acts);
  col_tile_issue((a.skip ? a.skip : a.acts) + (long long)b * a.h_bs, H, a.h_ld, t0, t_skip);
  col_tile_issue(a.z + (long long)b * a.z_bs, C, a.z_ld, t0, t_z);
  col_step_fill(s_rs, wave, a0, a1);
  float bb_rs[2][4], bb_end[2][4], bb_st[2][4];
  col_bias_issue(a.b_rs, s_rs.RT, wave, rq, bb_rs);
  col_bias_issue(a.b_end, s_end.RT, wave, rq, bb_end);
  col_bias_issue(a.b_st, s_st.RT, wave, rq, bb_st);
  // coupling items of this thread: channel group k = (tid >> 4) + 32 i (i < 2: half <= 128), column tid & 15
  const int ngroups = half / 2;
  float mb[2][4], ms[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = (tid >> 4) + 32 * i;
    const int c0 = 2 * (k < ngroups ? k : ngroups - 1);
    mb[i][0] = a.mix_bias[c0];
    mb[i][1] = a.mix_bias[c0 + 1];
    mb[i][2] = a.mix_bias[half + c0];
    mb[i][3] = a.mix_bias[half + c0 + 1];
    ms[i][0] = a.mix_scale[c0];
    ms[i][1] = a.mix_scale[c0 + 1];
    ms[i][2] = a.mix_scale[half + c0];
    ms[i][3] = a.mix_scale[half + c0 + 1];
  }
  float w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) w[i] = a.mix_w[i];
  COL_STAMP(6);
  col_tile_land(t_acts, true, H, KH, t0, L, R1);
  col_tile_land(t_skip, a.skip != nullptr, H, KH, t0, L, R2);
  col_tile_land(t_z, true, C, C, t0, L, R3);
  COL_STAMP(7);
  __syncthreads();
  COL_STAMP(1);

  col_floatx4 acc0, acc1;
  // ---- s = (W_rs acts + b) + skip
  col_step_run(s_rs, wave, R1, a0, a1, acc0, acc1);
  col_step_fill(s_end, wave, a0, a1);
  {
    const int nt = col_tiles_of(wave, s_rs.RT);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < nt) {
        const int r0 = (wave + 8 * i) * 16 + rq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* p = R2 + (r0 + r) * COL_T + col;
          *p = ((i ? acc1[r] : acc0[r]) + bb_rs[i][r]) + *p;
        }
      }
  }
  __syncthreads();
  COL_STAMP(2);
  // ---- m | logs = W_end s + b
  col_step_run(s_end, wave, R2, a0, a1, acc0, acc1);
  col_step_fill(s_st, wave, a0, a1);
  {
    const int nt = col_tiles_of(wave, s_end.RT);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < nt) {
        const int r0 = (wave + 8 * i) * 16 + rq;
#pragma unroll
        for (int r = 0; r < 4; ++r) R1[(r0 + r) * COL_T + col] = (i ? acc1[r] : acc0[r]) + bb_end[i][r];
      }
  }
  __syncthreads();
  COL_STAMP(3);
  // ---- coupling, InvConvNear^-1, ActNorm^-1 on the channel groups {2k, 2k+1, half+2k, half+2k+1}
  {
    float* zb = a.z + (long long)b * a.z_bs;
    const int n = tid & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = (tid >> 4) + 32 * i;
      if (k < ngroups) {
        const int c0 = 2 * k;
        const float m0 = R1[c0 * COL_T + n], m1 = R1[(c0 + 1) * COL_T + n];
        const float l0 = R1[(half + c0) * COL_T + n], l1 = R1[(half + c0 + 1) * COL_T + n];
        const float in0 = R3[c0 * COL_T + n], in1 = R3[(c0 + 1) * COL_T + n];
        const float in2 = (R3[(half + c0) * COL_T + n] - m0) * expf(-l0);
        const float in3 = (R3[(half + c0 + 1) * COL_T + n] - m1) * expf(-l1);
        float o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) o[m] = w[m * 4 + 0] * in0 + w[m * 4 + 1] * in1 + w[m * 4 + 2] * in2 + w[m * 4 + 3] * in3;
        const float y0 = (o[0] - mb[i][0]) * ms[i][0];
        const float y1 = (o[1] - mb[i][1]) * ms[i][1];
        const float y2 = (o[2] - mb[i][2]) * ms[i][2];
        const float y3 = (o[3] - mb[i][3]) * ms[i][3];
        const int t = t0 + n;
        if (t < L) {
          zb[(long long)c0 * a.z_ld + t] = y0;
          zb[(long long)(c0 + 1) * a.z_ld + t] = y1;
          zb[(long long)(half + c0) * a.z_ld + t] = y2;
          zb[(long long)(half + c0 + 1) * a.z_ld + t] = y3;
        }
        R2[c0 * COL_T + n] = y0;
        R2[(c0 + 1) * COL_T + n] = y1;
      }
    }
    for (int e = half * COL_T + tid; e < KZ * COL_T; e += 512) R2[e] = 0.f;  // K padding of the next step
  }
  if (!a.hnext) return;
  __syncthreads();
  COL_STAMP(4);
  // ---- h = W_start z0 + b
  col_step_run(s_st, wave, R2, a0, a1, acc0, acc1);
  {
    const int nt = col_tiles_of(wave, s_st.RT);
    float* hb = a.hnext + (long long)b * a.h_bs;
    const int t = t0 + col;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < nt) {
        const int r0 = (wave + 8 * i) * 16 + rq;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r0 + r < H && t < L) hb[(long long)(r0 + r) * a.h_ld + t] = (i ? acc1[r] : acc0[r]) + bb_st[i][r];
      }
  }
  COL_STAMP(5);
}

struct OprojLnArgs {
  const float* x;    // [B][H][ld]: attention output
  const float* res;  // [B][H][ld]: residual (the layer's input)
  float* y;          // [B][H][ld]: LayerNorm(res + conv_o(x)); may alias res
  long long bs;
  int ld;
  const int* len;
  int len_mul, len_const;
  const float *w, *b;  // pack_col16 of conv_o: H rows, K = H
  const float *gamma, *beta;
  int H;
  float eps;
};

__global__ __launch_bounds__(512) void oproj_ln_kernel(const OprojLnArgs a) {
  GLOW_PRIO();
  __shared__ float lds[2 * COL_MAXROWS * COL_T];
  __shared__ float red[32][COL_T + 1];
  float* R1 = lds;
  float* R2 = lds + COL_MAXROWS * COL_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int t0 = blockIdx.x * COL_T;
  if (t0 >= L) return;
  const int H = a.H;
  const int KH = (H + 15) & ~15;
  const ColStep st{reinterpret_cast<const float4*>(a.w), KH / 16, KH / 16};
  const int col = lane & 15, rq = 4 * (lane >> 4);
  // LayerNorm thread map: (n = tid & 15, g = tid >> 4) holds rows g, g + 32, ...
  const int n = tid & 15, g = tid >> 4;
  constexpr int NV = COL_MAXROWS / 32;
  // ---- all loads of the launch in one batch
  float4 a0[COL_RING], a1[COL_RING];
  ColTileRegs t_x, t_res;
  col_tile_issue(a.x + (long long)b * a.bs, H, a.ld, t0, t_x);
  col_tile_issue(a.res + (long long)b * a.bs, H, a.ld, t0, t_res);
  col_step_fill(st, wave, a0, a1);
  float bb[2][4];
  col_bias_issue(a.b, st.RT, wave, rq, bb);
  float gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = g + 32 * i;
    gm[i] = a.gamma[c < H ? c : H - 1];
    bt[i] = a.beta[c < H ? c : H - 1];
  }
  col_tile_land(t_x, true, H, KH, t0, L, R1);
  col_tile_land(t_res, true, H, KH, t0, L, R2);
  __syncthreads();
  col_floatx4 acc0, acc1;
  col_step_run(st, wave, R1, a0, a1, acc0, acc1);
  {
    const int nt = col_tiles_of(wave, st.RT);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (i < nt) {
        const int r0 = (wave + 8 * i) * 16 + rq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* p = R2 + (r0 + r) * COL_T + col;
          *p = ((i ? acc1[r] : acc0[r]) + bb[i][r]) + *p;
        }
      }
  }
  __syncthreads();
  // LayerNorm over the H rows of each column (glow_tts/attentions.py LayerNorm: mean / biased variance over channels)
  float v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = g + 32 * i;
    v[i] = c < H ? R2[c * COL_T + n] : 0.f;
    s += v[i];
  }
  red[g][n] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) mean += red[k][n];
  mean /= (float)H;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = (g + 32 * i < H) ? v[i] - mean : 0.f;
    q += d * d;
  }
  red[g][n] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) var += red[k][n];
  var /= (float)H;
  const float rstd = rsqrtf(var + a.eps);
  const int t = t0 + n;
  if (t >= L) return;
  float* yb = a.y + (long long)b * a.bs + t;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = g + 32 * i;
    if (c < H) yb[(long long)c * a.ld] = (v[i] - mean) * rstd * gm[i] + bt[i];
  }
}

}  // namespace mi355tts
