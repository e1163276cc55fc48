// A whole multi-receptive-field stage of HiFi-GAN in ONE launch, for the narrow stages (C = 8 / 16 channels):
//
//   y = ( rb_{K0}(x) + rb_{K1}(x) + rb_{K2}(x) ) / 3                          hifi_gan/models.py:191-197
//   rb_K(x): for d in dilations:  x = x + conv2_{K,1}( lrelu( conv1_{K,d}( lrelu(x) ) ) )      :91-98
//
// These stages ('medium': 16 and 8 channels at 128 / 256 samples per mel frame) hold a third of the vocoder's
// FLOPs but are far too narrow for the 32-row / 64-channel tiles of conv_mfma.h (4x the rows and up to 8x the
// K-depth of MFMA work would be zeros), and un-fused they move 18 x 2 planes through the caches per stage at
// 6 FLOP/B.  Here a workgroup owns T output columns of one batch row and runs whole chains on an LDS-resident
// tile: the input is staged with 64 columns of halo on either side (the three chains need 12 / 36 / 60), every
// intermediate stays in LDS / registers, and only chain SUMS are written: two workgroups per tile, one for the
// k = 11 chain (11/21 of the work) and one for the k = 3 + k = 7 chains (10/21), each writing one [C x T] tile;
// the consumer adds the two planes and divides by 3 as it loads them.  Cache traffic per stage = two reads + two
// writes of the plane (+ 2 x 64 / T halo) instead of 36.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact f32, bit-equal to an fmaf chain) — M = 16 output channels (all of
// them; C = 8 uses half the rows), N = 16 time columns, K-dim = 4 input channels of one tap.  A = weights,
// pre-packed per (tap, channel quad) as one dword per lane and streamed from L2; B = activations straight out of
// LDS, one conflict-free ds_read_b32 per MFMA (row stride = 16 mod 32 floats).  The leaky-ReLU is applied when a
// value is WRITTEN to LDS (once per element), the raw residual stream lives in registers in the C/D layout
// (the same lane owns the same (channel, column) positions in every conv of a chain).
//
// A conv only computes the columns later convs still need (the halo shrinks by (K-1)/2*d per conv); the
// 16-column blocks of the tile are dealt to the waves so that every wave always owns its share of the core
// blocks and the halo blocks nearest to the core first — the active set of a wave is a prefix of its slots and
// the MFMA loop is instantiated per slot count (no predication inside the loop).
#pragma once
#include <hip/hip_runtime.h>

#include "conv_mfma.h"

namespace mi355tts {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int MRF_HALO = 64;      // staged columns on either side of the tile (>= the deepest chain's receptive half-width)
constexpr int MRF_MAX_STEPS = 3;  // dilation steps per chain
constexpr int MRF_TAB_DIL = 3 * MRF_MAX_STEPS * 2;
constexpr int MRF_TAB_INTS = MRF_TAB_DIL + 3 * MRF_MAX_STEPS;

// probe builds (tools/probe/mrf_bench.hip -DMRF_ABL=bits; results are WRONG when set): 1 = no B operand reads in the MFMA
// loop, 2 = no barriers, 4 = epilogue LDS stores skipped at run time, 8 = no A fragment loads in the loop, 32 = MFMA loops skipped at run time
#ifndef MRF_ABL
#define MRF_ABL 0
#endif

struct MrfArgs {
  const float* x;  // stage input [B][C][ld] (the upsampler's output)
  float* y;        // out: rb_K0(x) + rb_K1(x)  [B][C][ld]
  float* y2;       // out: rb_K2(x), same geometry; the consumer forms (y + y2) / 3
  long long bs;
  int ld;
  const int* len;  // valid length of row b = len ? len[b] * len_mul : len_const
  int len_mul;
  int len_const;
  const float* w;     // packed A fragments, see pack_mrf_conv (weights_pack.h)
  const float* bias;  // [chain][step][conv][16], zero padded
  // device table (indexed with run-time chain / step: by-value kernel-argument arrays would be copied to scratch):
  //   tab[(chain * MRF_MAX_STEPS + step) * 2 + conv] = float offset of that conv's fragments in w,
  //   tab[MRF_TAB_DIL + chain * MRF_MAX_STEPS + step] = conv1's dilation of that step (conv2 has dilation 1)
  const int* tab;
  int nsteps;
  float slope;
};

// acc[s] += conv taps over one staged source for the wave's first NB slots.
//   an   : tap 0's A fragments, already requested by the caller (a conv's first weights are cold in L1: they are
//          asked for while the PREVIOUS conv's epilogue runs)
//   wp   : this conv's fragments + lane          ([tap][C/4][64] floats)
//   src  : LDS source ([C][W], lrelu already applied)
//   boff : per slot, (lane >> 4) * W + 16 * block + (lane & 15)  (the B element of tap offset 0, channel quad 0)
//   t0   : tap 0's column offset (-pad);  taps are `dil` columns apart
template <int C, int W, int NB, int NS, int CORE, int NW>
__device__ __forceinline__ void mrf_conv_taps(floatx4 (&acc)[NS], float (&an)[C / 4], const float* __restrict__ wp,
                                              const float* __restrict__ src, const int core_off, const int (&halo_off)[NS - CORE],
                                              const int t0, const int dil, const int K) {
  constexpr int CQ = C / 4;
  // One step = one (tap, channel quad) = NB MFMAs, one per slot.  Software pipeline, written out: the B operands of
  // the next step (NB ds_read_b32) and the A fragment of the same quad one tap ahead (one global load) are requested
  // in the shadow of this step's MFMAs — one request pinned behind each MFMA — and nothing moves across a step
  // boundary, which bounds the live registers to two steps' operands.  The tap loop is a real loop (K is a run-time
  // value): unrolled over (K, NB) the kernel was ~100 KB of straight-line code.  On the last tap the prefetches read
  // one tap too far: the next conv's fragments (inside the weight arena) and columns inside the LDS slack — values
  // nothing uses.  The core slots of a wave are 16*NW columns apart: ONE address register + immediate offsets.
  const float* pc = src + core_off + t0;  // core slot s reads pc[s * 16 * NW + q * 4 * W]
  const float* ph[NS - CORE];
#pragma unroll
  for (int h = 0; h < NS - CORE; ++h) ph[h] = src + halo_off[h] + t0;
  auto bread = [&](int s, int q) -> float { return s < CORE ? pc[s * 16 * NW + q * 4 * W] : ph[s - CORE][q * 4 * W]; };
  float bcur[NB], bnxt[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) bcur[s] = bread(s, 0);
  const float* wt = wp + CQ * 64;  // the next tap's fragments
  const int taps = ((MRF_ABL & 32) && dil < 99) ? 0 : K;
#pragma unroll 1
  for (int tap = 0; tap < taps; ++tap) {
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const float av = an[q];
      if (!(MRF_ABL & 8)) an[q] = wt[q * 64];
      if (MRF_ABL & 64) {
#pragma unroll
        for (int s = 0; s < NB; ++s) bnxt[s] = bcur[s];
      } else if (MRF_ABL & 1) {
#pragma unroll
        for (int s = 0; s < NB; ++s) bnxt[s] = bcur[s] + 1.0f;
      } else if (q + 1 < CQ) {
#pragma unroll
        for (int s = 0; s < NB; ++s) bnxt[s] = bread(s, q + 1);
      } else {
        pc += dil;
#pragma unroll
        for (int h = 0; h < NS - CORE; ++h) ph[h] += dil;
#pragma unroll
        for (int s = 0; s < NB; ++s) bnxt[s] = bread(s, 0);
      }
#pragma unroll
      for (int s = 0; s < NB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bcur[s], acc[s], 0, 0, 0);
      // issue order (0x008 = MFMA, 0x020 = VMEM read, 0x100 = LDS read)
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (s == 0 && !(MRF_ABL & 8)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (!(MRF_ABL & 65)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < NB; ++s) bcur[s] = bnxt[s];
    }
    wt += CQ * 64;
  }
}

// A value the optimiser must treat as new (wave-uniform ints): the second chain of a workgroup recomputes its staging
// addresses instead of keeping the first chain's six 64-bit pointers alive — spilled to scratch — through a whole chain.
__device__ __forceinline__ int mrf_opaque(int v) {
#if defined(__AMDGCN__)
  asm volatile("" : "+s"(v));
#endif
  return v;
}

template <int C, int T, int NW>
struct MrfGeom {
  static constexpr int W = T + 2 * MRF_HALO + 16;  // LDS row stride: = 16 (mod 32) floats -> B reads of 4 rows x 16 columns hit 64 distinct banks
  static constexpr int NCOL = T + 2 * MRF_HALO;    // staged columns
  static constexpr int CORE = T / 16 / NW;         // core slots per wave
  static constexpr int HS = (2 * MRF_HALO / 16 + NW - 1) / NW;  // halo slots per wave
  static constexpr int NS = CORE + HS;
  // two planes (conv1's and conv2's operands) + slack on either side: edge blocks of a conv read up to
  // ((K-1)/2 + 1)*d columns before / past the staged rows (values only garbage columns use)
  static constexpr int LDS_FLOATS = 64 + 2 * C * W + 64;
  static_assert(T % (16 * NW) == 0 && W % 32 == 16 && HS >= 1 && HS <= 2, "tile geometry");
  static_assert(C == 8 || C == 16, "one 16-row MFMA block of output channels");
};

// grid = (tiles, 2, B).  blockIdx.y = 0: the K2 chain alone -> y2;  1: the K0 then the K1 chain, summed -> y.
// The consumer forms (y + y2) / 3 = ((rb_K0 + rb_K1) + rb_K2) / 3 — the reference's summation order — when it loads
// its input (ConvArgs::x2 / in_div).  Two workgroup kinds of 10/21 and 11/21 of a tile's work instead of one
// workgroup per tile: twice the workgroups of half the duration (a launch is only 1.2 - 4 tiles per CU deep).
#ifndef MRF_MIX
#define MRF_MIX 256
#endif
#ifndef MRF_OCC
#define MRF_OCC(T, NW) ((T) / (NW) <= 64 ? 3 : 2)  // waves per SIMD the register allocation aims at (tools/probe/mrf_bench.hip overrides)
#endif
template <int C, int T, int NW, int K0, int K1, int K2>
__global__ __launch_bounds__(64 * NW, MRF_OCC(T, NW)) void mrf_small_kernel(const MrfArgs a) {
  using G = MrfGeom<C, T, NW>;
  constexpr int W = G::W, CORE = G::CORE, HS = G::HS, NS = G::NS, CQ = C / 4;
  constexpr int NT = 64 * NW;
  __shared__ float lds[G::LDS_FLOATS];
  float* const XL = lds + 64;     // lrelu(current x of the running chain): conv1's operand
  float* const TB = XL + C * W;   // lrelu(conv1 + bias): conv2's operand

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.z;
  int tile_x, tile_y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int gx = gridDim.z > 1 ? row_tiles(L, T) : (int)gridDim.x / 2;  // ragged batch: this row's own tiles only (conv_mfma.h)
  // 1-D grid of 2 * tiles workgroups: runs of MRF_MIX "k = 11" workgroups alternate with runs of MRF_MIX "k = 3 + 7"
  // ones, so that the workgroups that end up sharing a CU (dispatch order: one per CU, then the second, ...) are of
  // both kinds and at different points of their conv sequence — workgroups of one kind started together run their
  // MFMA loops and their epilogues in lockstep and cannot cover for each other.
  const int lin = blockIdx.x;
  if (lin >= 2 * gx) return;
  const int chunk = lin / (2 * MRF_MIX), r = lin - chunk * (2 * MRF_MIX);
  const int n_in = gx - chunk * MRF_MIX < MRF_MIX ? gx - chunk * MRF_MIX : MRF_MIX;
  const int part = r / n_in;
  xcd_tile_lin(chunk * MRF_MIX + (r - part * n_in), gx, 1, tile_x, tile_y);  // neighbouring tiles (and a tile's two parts) share their input through one XCD's L2
  const int j0 = tile_x * T;
  if (j0 >= L) return;
  const int gx0 = j0 - MRF_HALO;  // global column of LDS column 0
  const float slope = a.slope;
  const float* xb = a.x + (long long)b * a.bs;

  // ---- this wave's slots: CORE core blocks (16*NW columns apart, from block 4 + wave), then its halo blocks, nearest
  // to the core first: halo entry e = wave + NW*h: even -> left block 3 - e/2, odd -> right block 4 + T/16 + e/2.
  // A lane owns, in the C/D layout, rows 4*rq + r (r = 0..3) of column col_of(s) of each slot.
  const int colq = lane & 15, rq = lane >> 4;
  const int core_col = 16 * (MRF_HALO / 16 + wave) + colq;
  int halo_col[HS];
#pragma unroll
  for (int h = 0; h < HS; ++h) {
    const int e = wave + NW * h;
    halo_col[h] = 16 * ((e & 1) ? (MRF_HALO / 16 + T / 16 + (e >> 1)) : (MRF_HALO / 16 - 1 - (e >> 1))) + colq;
  }
  auto col_of = [&](int s) -> int { return s < CORE ? core_col + 16 * NW * s : halo_col[s - CORE]; };
  const int core_off = rq * W + core_col;  // B element (tap offset 0, channel quad 0) of core slot 0
  int halo_off[HS];
#pragma unroll
  for (int h = 0; h < HS; ++h) halo_off[h] = rq * W + halo_col[h];
  const int row0 = 4 * rq;
  const bool rows_ok = row0 < C;  // C = 8: the upper half of the MFMA block is padding
  auto inside = [&](int s) -> bool {  // the slot's column lies inside the sequence
    const int g = gx0 + col_of(s);
    return g >= 0 && g < L;
  };

  floatx4 xres[NS];
  floatx4 acc[NS];
  float an[CQ];  // tap 0's A fragments of the NEXT conv, requested one epilogue ahead
  float bn[4];   // its bias

  // chain start: XL = lrelu(x) for the whole tile (16-byte loads, branch-free, zero outside the sequence) and the raw
  // residual stream of the owned positions straight from global memory (L2-hot: the tile was just read)
  auto stage = [&](const int gx0) __attribute__((always_inline)) {  // (shadows gx0 with the caller's — possibly laundered — copy)
    constexpr int F4 = G::NCOL / 4;
    constexpr int NF4 = C * F4;
    constexpr int NE = (NF4 + NT - 1) / NT;
    const int ld_last4 = a.ld - 4;
    float4 pre[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / F4, f = e - row * F4;
      const int c0 = gx0 + 4 * f;
      pre[i] = *reinterpret_cast<const float4*>(xb + (row < C ? row : C - 1) * a.ld + (c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0)));
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / F4, f = e - row * F4;
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
      if (e < NF4) *reinterpret_cast<float4*>(XL + row * W + 4 * f) = v;
    }
    // (after the tile is in LDS: its staging registers are dead by now; first needed at the first conv2's epilogue)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float* xp = xb + (inside(s) ? gx0 + col_of(s) : 0) + (rows_ok ? row0 : 0) * a.ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) xres[s][r] = xp[r * a.ld];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) xres[s][r] = (inside(s) && rows_ok) ? xres[s][r] : 0.f;
  };
  auto prefetch = [&](int chain, int step, int cv) __attribute__((always_inline)) {  // tap 0's fragments and the bias of conv (chain, step, cv)
    const float* wp = a.w + a.tab[(chain * MRF_MAX_STEPS + step) * 2 + cv] + lane;
#pragma unroll
    for (int q = 0; q < CQ; ++q) an[q] = wp[q * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) bn[r] = a.bias[((chain * MRF_MAX_STEPS + step) * 2 + cv) * 16 + row0 + r];
  };

  // one ResBlock1 chain; `first` = it starts this workgroup's sum, `next_chain` >= 0 = the chain that follows (its first
  // weights are requested during this chain's last epilogue)
  auto run_chain = [&](const int K, const int chain, const bool first, const int next_chain) __attribute__((always_inline)) {
    const int P2 = (K - 1) / 2;
    // remaining halo after each conv of this chain (what later convs still need on either side)
    int need = 0;
    for (int s = 0; s < a.nsteps; ++s) need += P2 * (a.tab[MRF_TAB_DIL + chain * MRF_MAX_STEPS + s] + 1);
    // (no barrier before re-staging XL: the previous chain's conv1s — XL's only readers — all ended on a barrier,
    //  and its last conv2 reads TB only)
    stage(first ? gx0 : mrf_opaque(gx0));
    __syncthreads();
    for (int step = 0; step < a.nsteps; ++step) {
      const int dil = a.tab[MRF_TAB_DIL + chain * MRF_MAX_STEPS + step];
      const bool last = step == a.nsteps - 1;
#pragma unroll
      for (int cv = 0; cv < 2; ++cv) {
        const int d = cv == 0 ? dil : 1;
        need -= P2 * d;  // halo the LATER convs still need = half-width of this conv's output range beyond the core
        // halo slots of this wave inside the range: entries e < 2 * ceil(need / 16)
        const int n_act = 2 * ((need + 15) >> 4);
        int nh = 0;
#pragma unroll
        for (int h = 0; h < HS; ++h) nh += (wave + NW * h) < n_act ? 1 : 0;
        const float* wp = a.w + a.tab[(chain * MRF_MAX_STEPS + step) * 2 + cv] + lane;
        const float* src = cv == 0 ? XL : TB;
        float bb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bb[r] = bn[r];
        // accumulators start at the bias (one VALU op less per output than adding it in the epilogue)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = floatx4{bb[0], bb[1], bb[2], bb[3]};
        // wave-uniform slot count -> one instantiation of the MFMA loop per count
        if (HS >= 2 && nh >= 2) mrf_conv_taps<C, W, (HS >= 2 ? CORE + 2 : NS), NS, CORE, NW>(acc, an, wp, src, core_off, halo_off, -P2 * d, d, K);
        else if (nh >= 1) mrf_conv_taps<C, W, CORE + 1, NS, CORE, NW>(acc, an, wp, src, core_off, halo_off, -P2 * d, d, K);
        else mrf_conv_taps<C, W, CORE, NS, CORE, NW>(acc, an, wp, src, core_off, halo_off, -P2 * d, d, K);
        const int nb = CORE + nh;
        // the next conv's first weights + bias go out before this epilogue
        if (cv == 0) prefetch(chain, step, 1);
        else if (!last) prefetch(chain, step + 1, 0);
        else if (next_chain >= 0) prefetch(next_chain, 0, 0);
        if (cv == 0) {
          // TB = lrelu(conv1 + bias), zero outside the sequence (conv2's zero padding)
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (s < nb && rows_ok) {
              const bool in = inside(s);
              float* tp = TB + row0 * W + col_of(s);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float v = acc[s][r];
                v = v > 0.f ? v : v * slope;
                if (!(MRF_ABL & 4) || a.nsteps == 99) tp[r * W] = in ? v : 0.f;
              }
            }
          }
        } else {
          // x = x + conv2 + bias; XL = lrelu(x) for the next step
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (s < nb) {
              const bool in = inside(s);
              float* xp = XL + row0 * W + col_of(s);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = in ? acc[s][r] + xres[s][r] : 0.f;
                xres[s][r] = v;
                if (!last && rows_ok && (!(MRF_ABL & 4) || a.nsteps == 99)) xp[r * W] = v > 0.f ? v : v * slope;
              }
            }
          }
        }
        if (!(last && cv == 1) && !(MRF_ABL & 2)) __syncthreads();
      }
    }
    // the chain's result goes to this workgroup's output plane, in the reference's summation order (xs = rb0; xs += rb1):
    // the first chain of a workgroup stores, the second adds to what the same lanes stored (a register-resident sum
    // would hold 16 more VGPRs through the whole second chain — the difference between spilling and not at 3 per CU)
    if (rows_ok) {
      float* yb = (part == 0 ? a.y2 : a.y) + (long long)b * a.bs;
#pragma unroll
      for (int s = 0; s < CORE; ++s) {
        const int g = gx0 + col_of(s);
        if (g < L) {
          float* yp = yb + (long long)row0 * a.ld + g;
          if (first) {
#pragma unroll
            for (int r = 0; r < 4; ++r) yp[(long long)r * a.ld] = xres[s][r];
          } else {
            float prev[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) prev[r] = yp[(long long)r * a.ld];
#pragma unroll
            for (int r = 0; r < 4; ++r) yp[(long long)r * a.ld] = prev[r] + xres[s][r];
          }
        }
      }
    }
  };
  if (part == 0) {
    prefetch(2, 0, 0);
    run_chain(K2, 2, true, -1);
  } else {
    prefetch(0, 0, 0);
    run_chain(K0, 0, true, 1);
    run_chain(K1, 1, false, -1);
  }
}


// =====================================================================================================================
// The 8-channel stage on v_mfma_f32_4x4x1_16B_f32 (round 3): with C = 8 the 16-row MFMA above runs half its rows on
// padding.  The 16-block 4 x 4 x 1 form with CBSZ = 4 broadcasts ONE block's A to all sixteen: D[i][lane] += A[4*ABID + i] *
// B[lane], i = 0..3 — a 4-row x 64-column rank-1 update, two of which (ABID = 2 ci, 2 ci + 1) cover the 8 output channels of
// 64 columns for one (input channel, tap) at the full f32 matrix rate with no padding at all:
//   A: ONE register per tap holds W[co][ci][tap] for all 8 x 8 (co, ci) pairs — lane = 8 ci + co — and ABID picks the input
//      channel: one coalesced 256-byte load per tap feeds 16 MFMAs per 64-column block;
//   B: lane l holds x[ci][c0 + l + tap * dil]: 64 consecutive floats of one LDS row, one conflict-free ds_read_b32;
//   D: two float4 per 64-column block: a lane owns all 8 channels of ONE column.
// Blocks are 64 columns on the tile's own grid: [left halo | T / 64 core blocks | right halo].  Two waves per workgroup: wave 0
// owns the left halo block and the first half of the core, wave 1 the second half and the right halo block — every conv but
// a chain's last runs T / 128 + 1 blocks on each wave (balanced; the last one the core only), a lane keeps its columns from
// conv to conv, so the raw residual stream stays in registers and LDS holds two planes (26 KB: six workgroups per CU).
template <int T, int NB>
__device__ __forceinline__ void mrf8_conv_taps(floatx4 (&acc)[T / 128 + 1][2], float& an, const float* __restrict__ wp,
                                               const float* __restrict__ pcore, const float* __restrict__ phalo, const int dil, const int K) {
  constexpr int W = T + 2 * MRF_HALO + 16;
  constexpr int CPW = T / 128;  // core blocks per wave; slot CPW (when NB > CPW) is the wave's halo block
  // one step = one input channel of one tap = 2 * NB MFMAs (8 cycles each); the next step's B reads are pinned behind them,
  // operands alternate between two register sets (no moves)
  auto bread = [&](int s, int ci) -> float { return s < CPW ? pcore[ci * W + 64 * s] : phalo[ci * W]; };
  float b0[NB], b1[NB];
#pragma unroll
  for (int s = 0; s < NB; ++s) b0[s] = bread(s, 0);
  const float* wt = wp + 64;  // the next tap's fragment
#pragma unroll 1
  for (int tap = 0; tap < K; ++tap) {
    const float av = an;
    an = wt[0];  // (reads one tap past the conv on the last iteration: the next conv's fragment or the arena's slack)
    wt += 64;
    // (ABID must be a literal: the eight input channels are spelled out)
#define MRF8_STEP(ci, BC, BN)                                                                      \
  {                                                                                                \
    if (ci + 1 < 8) {                                                                              \
      _Pragma("unroll") for (int s = 0; s < NB; ++s) BN[s] = bread(s, ci + 1);                     \
    } else {                                                                                       \
      pcore += dil;                                                                                \
      phalo += dil;                                                                                \
      _Pragma("unroll") for (int s = 0; s < NB; ++s) BN[s] = bread(s, 0);                          \
    }                                                                                              \
    _Pragma("unroll") for (int s = 0; s < NB; ++s) {                                               \
      acc[s][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, BC[s], acc[s][0], 4, 2 * ci, 0);          \
      acc[s][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, BC[s], acc[s][1], 4, 2 * ci + 1, 0);      \
    }                                                                                              \
    _Pragma("unroll") for (int s = 0; s < NB; ++s) {                                               \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
      if (s == 0 && ci == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
    MRF8_STEP(0, b0, b1) MRF8_STEP(1, b1, b0) MRF8_STEP(2, b0, b1) MRF8_STEP(3, b1, b0)
    MRF8_STEP(4, b0, b1) MRF8_STEP(5, b1, b0) MRF8_STEP(6, b0, b1) MRF8_STEP(7, b1, b0)
#undef MRF8_STEP
  }
}

// grid = (2 * tiles, 1, B): workgroup kinds as in mrf_small_kernel (part 0: the K2 chain -> y2; part 1: K0 then K1 -> y).
// a.w / a.tab hold the 4x4x1 packing (pack_mrf8_conv): per conv [tap][64 lanes], lane = 8 * ci + co.
template <int T, int K0, int K1, int K2>
__global__ __launch_bounds__(128) void mrf8_kernel(const MrfArgs a) {
  constexpr int C = 8, W = T + 2 * MRF_HALO + 16, NCOL = T + 2 * MRF_HALO;
  constexpr int CPW = T / 128, NS = CPW + 1;  // slots per wave: its core blocks, then its halo block
  constexpr int NT = 128;
  static_assert(T % 128 == 0, "two waves split the core blocks evenly");
  __shared__ float lds[64 + 2 * C * W + 128];
  float* const XL = lds + 64;     // lrelu(current x): conv1's operand
  float* const TB = XL + C * W;   // lrelu(conv1 + bias): conv2's operand

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int b = blockIdx.z;
  int tile_x, tile_y;
  const int L = a.len ? a.len[b] * a.len_mul : a.len_const;
  const int gx = gridDim.z > 1 ? row_tiles(L, T) : (int)gridDim.x / 2;
  const int lin = blockIdx.x;
  if (lin >= 2 * gx) return;
  const int chunk = lin / (2 * MRF_MIX), rr = lin - chunk * (2 * MRF_MIX);
  const int n_in = gx - chunk * MRF_MIX < MRF_MIX ? gx - chunk * MRF_MIX : MRF_MIX;
  const int part = rr / n_in;
  xcd_tile_lin(chunk * MRF_MIX + (rr - part * n_in), gx, 1, tile_x, tile_y);
  const int j0 = tile_x * T;
  if (j0 >= L) return;
  const int gx0 = j0 - MRF_HALO;
  const float slope = a.slope;
  const float* xb = a.x + (long long)b * a.bs;

  // this lane's LDS column in each slot: core blocks 1 + wave * CPW + s, the halo block 0 (wave 0) / 1 + T / 64 (wave 1)
  const int core_col = 64 * (1 + wave * CPW) + lane;
  const int halo_col = (wave == 0 ? 0 : 64 * (1 + T / 64)) + lane;
  auto col_of = [&](int s) -> int { return s < CPW ? core_col + 64 * s : halo_col; };
  auto inside = [&](int s) -> bool {
    const int g = gx0 + col_of(s);
    return g >= 0 && g < L;
  };

  floatx4 acc[NS][2];
  float xres[NS][8];  // the raw residual stream of the columns this lane owns
  float an;           // tap 0's fragment of the NEXT conv
  float bn[8];        // its bias

  auto stage = [&](const int gx0) __attribute__((always_inline)) {
    constexpr int F4 = NCOL / 4;
    constexpr int NF4 = C * F4;
    constexpr int NE = (NF4 + NT - 1) / NT;
    const int ld_last4 = a.ld - 4;
    float4 pre[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / F4, f = e - row * F4;
      const int c0 = gx0 + 4 * f;
      pre[i] = *reinterpret_cast<const float4*>(xb + (row < C ? row : C - 1) * a.ld + (c0 < 0 ? 0 : (c0 > ld_last4 ? ld_last4 : c0)));
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + NT * i;
      const int row = e / F4, f = e - row * F4;
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
      if (e < NF4) *reinterpret_cast<float4*>(XL + row * W + 4 * f) = v;
    }
    // the raw residual stream of the owned columns, straight from global memory (L2-hot): 256-byte rows per block
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bool in = inside(s);
      const float* xp = xb + (in ? gx0 + col_of(s) : 0);
#pragma unroll
      for (int r = 0; r < 8; ++r) xres[s][r] = xp[(long long)r * a.ld];
#pragma unroll
      for (int r = 0; r < 8; ++r) xres[s][r] = in ? xres[s][r] : 0.f;
    }
  };
  auto prefetch = [&](int chain, int step, int cv) __attribute__((always_inline)) {
    an = a.w[a.tab[(chain * MRF_MAX_STEPS + step) * 2 + cv] + lane];
#pragma unroll
    for (int r = 0; r < 8; ++r) bn[r] = a.bias[((chain * MRF_MAX_STEPS + step) * 2 + cv) * 16 + r];
  };

  auto run_chain = [&](const int K, const int chain, const bool first, const int next_chain) __attribute__((always_inline)) {
    const int P2 = (K - 1) / 2;
    int need = 0;
    for (int s = 0; s < a.nsteps; ++s) need += P2 * (a.tab[MRF_TAB_DIL + chain * MRF_MAX_STEPS + s] + 1);
    // (no barrier before re-staging XL: the previous chain's conv1s — XL's only readers — all ended on a barrier)
    stage(first ? gx0 : mrf_opaque(gx0));
    __syncthreads();
    for (int step = 0; step < a.nsteps; ++step) {
      const int dil = a.tab[MRF_TAB_DIL + chain * MRF_MAX_STEPS + step];
      const bool last = step == a.nsteps - 1;
#pragma unroll
      for (int cv = 0; cv < 2; ++cv) {
        const int d = cv == 0 ? dil : 1;
        need -= P2 * d;  // what the LATER convs still need beyond the core on either side: > 0 -> the halo blocks run too
        const float* wp = a.w + a.tab[(chain * MRF_MAX_STEPS + step) * 2 + cv] + lane;
        const float* src = cv == 0 ? XL : TB;
        float bb[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) bb[r] = bn[r];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc[s][0] = floatx4{bb[0], bb[1], bb[2], bb[3]};
          acc[s][1] = floatx4{bb[4], bb[5], bb[6], bb[7]};
        }
        const int nb = need > 0 ? NS : CPW;
        if (need > 0) mrf8_conv_taps<T, NS>(acc, an, wp, src + core_col - P2 * d, src + halo_col - P2 * d, d, K);
        else mrf8_conv_taps<T, CPW>(acc, an, wp, src + core_col - P2 * d, src + halo_col - P2 * d, d, K);
        if (cv == 0) prefetch(chain, step, 1);
        else if (!last) prefetch(chain, step + 1, 0);
        else if (next_chain >= 0) prefetch(next_chain, 0, 0);
        const bool final_conv = last && cv == 1;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (s < nb) {
            const bool in = inside(s);
            if (cv == 0) {
              float* tp = TB + col_of(s);
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                float v = acc[s][r >> 2][r & 3];
                v = v > 0.f ? v : v * slope;
                tp[r * W] = in ? v : 0.f;
              }
            } else {
              float* xp = XL + col_of(s);
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const float v = in ? acc[s][r >> 2][r & 3] + xres[s][r] : 0.f;
                xres[s][r] = v;
                if (!last) xp[r * W] = v > 0.f ? v : v * slope;
              }
            }
          }
        }
        if (!final_conv) __syncthreads();
      }
    }
    // the chain's result (core columns) goes to this workgroup's plane: the first chain stores, the second adds
    {
      float* yb = (part == 0 ? a.y2 : a.y) + (long long)b * a.bs;
#pragma unroll
      for (int s = 0; s < CPW; ++s) {
        const int g = gx0 + col_of(s);
        if (g < L) {
          float* yp = yb + g;
          if (first) {
#pragma unroll
            for (int r = 0; r < 8; ++r) yp[(long long)r * a.ld] = xres[s][r];
          } else {
            float prev[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) prev[r] = yp[(long long)r * a.ld];
#pragma unroll
            for (int r = 0; r < 8; ++r) yp[(long long)r * a.ld] = prev[r] + xres[s][r];
          }
        }
      }
    }
  };
  if (part == 0) {
    prefetch(2, 0, 0);
    run_chain(K2, 2, true, -1);
  } else {
    prefetch(0, 0, 0);
    run_chain(K0, 0, true, 1);
    run_chain(K1, 1, false, -1);
  }
}

}  // namespace mi355tts
